"""The reference's own operating point (n = 20 landmarks) at Monte-Carlo scale: B filters x T steps of the configs[0]
workload, inputs simulated on the device, as one LDS-resident launch (k_pool_run_known) vs the per-step replay.
usage: python tools/small_pool_bench.py [B] [T] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = synth.config1(steps=T)
cfg.filters, cfg.n = B, n
world = synth.make_world(n, cfg.half_extent, cfg.min_spacing, cfg.seed)
bt = capi.BatchEKF(B, n)
bt.simulate_known_log(cfg, world, vmax=min(n, 20))
for small, Trun in ((True, T), (False, min(T, 100))):
    bt.reset(); bt.set_small_map_path(small)
    bt.run_known(0, 1)
    st = bt.run_known(1, Trun, time_kernels=True)
    el = st["elapsed_ms"] * 1e-3
    print(f"n={n} B={B} steps={Trun - 1} LDS-resident={small}: {st['filter_steps'] / el / 1e6:8.2f} M filter steps/s, "
          f"{st['corrections'] / el / 1e6:8.2f} M corrections/s, elapsed {st['elapsed_ms']:.1f} ms, launches {st['rank2_launches']}", flush=True)
    if Trun == T:
        print("   mc:", {k: round(float(v), 4) for k, v in bt.mc_stats(T - 1).items()}, flush=True)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import binding as oracle
import copy
Bc = min(B, 64)
tw, li, zz, ii, _ = bt.download_log(want_truth=False)
c2 = copy.copy(cfg); c2.filters = Bc
sub = synth.KnownLog(c2, world, tw[:, :Bc], li[:, :Bc], zz[:, :Bc], ii[:Bc])
stt, _, info = oracle.batch_run_known(sub, oracle.STRUCTURED, nthreads=16)
print(f"cpu checker (structured, {info['threads']} threads, {Bc} filters): {Bc * T / info['seconds'] / 1e6:.3f} M filter steps/s, "
      f"{info['corrections'] / info['seconds'] / 1e6:.3f} M corrections/s")
