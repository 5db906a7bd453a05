"""Unknown association against a fully surveyed map (bench.py's unknown_association_large_prefix leg): B filters, n = 1000,
known_count = n; sweep of the covariance pass's tuning (rows per workgroup, rows per load/store group).
usage: python tools/large_prefix_bench.py [B] [steps] [rows,group ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
variants = [tuple(int(x) for x in v.split(",")) for v in sys.argv[3:]] or [(0, 0, 0)]   # rows, group[, delayed k]
n, J, W = 1000, 8, 3
Tu = 1 + W + K
lb = capi.BatchEKF(B, n)
world = synth.make_world(n, 12.0, 0.6, 3)
rng = np.random.default_rng(1000)
vm = 64
Ta = 1 + (n + vm - 1) // vm
a_lm = np.full((Ta, B, vm), -1, dtype=np.int32)
a_z = np.zeros((Ta, B, vm, 2))
for t in range(1, Ta):
    idx = np.arange((t - 1) * vm, min(n, t * vm))
    a_lm[t, :, :len(idx)] = idx
    a_z[t, :, :len(idx)] = world[idx][None] + rng.normal(0.0, 0.005, size=(B, len(idx), 2))
a_init = (world[None] + rng.normal(0.0, 0.005, size=(B, n, 2))).reshape(B, 2 * n)
lb.upload_known_log(np.zeros((Ta, B, 2)), a_lm, a_z, a_init)
lb.run_known(0, Ta)
lb.set_known_counts(n)
cfg = synth.config3(steps=Tu)
cfg.filters, cfg.n = B, n
lb.simulate_unknown_log(cfg, world, jmax=J)
lb.run_unknown(0, 1 + W)
t_next = 1 + W
for var in variants:
    rows, group = var[0], var[1]
    kd = var[2] if len(var) > 2 else 0
    steps = 3 if kd == 0 else 2 * max(1, kd // J)
    lb.set_tuning(rows, -1, group)
    lb.set_update_mode(kd)
    if t_next + steps > Tu:
        break
    st = lb.run_unknown(t_next, t_next + steps, time_kernels=True)
    t_next += steps
    N = 3 + 2 * n
    pass_ms = st["rank2_ms"] / st["rank2_launches"]
    print(f"rows={rows:3d} group={group:2d} delayed k={kd:2d}: step {st['elapsed_ms'] / steps:.3f} ms  pass/flush {pass_ms:.3f} ms x {st['rank2_launches']}"
          f" = {B * 16.0 * N * N / pass_ms / 1e6:.0f} GB/s  ({B * steps / st['elapsed_ms'] * 1e3:.0f} filter steps/s)", flush=True)
lb.close()
