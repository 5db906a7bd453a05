"""Delayed rank-2k update throughput on the GPU box (configs[4] share): k sweep, paired gain steps on / off,
symmetric gather on / off.  usage: python tools/delayed_bench.py [B=4096]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekf_slam_ml_amd import capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K, W = 120, 2
cfg = synth.config5(filters=B, steps=1 + W + K, n=1000)
bt = capi.BatchEKF(B, 1000)
bt.simulate_known_log(cfg, synth.make_world(1000, cfg.half_extent, cfg.min_spacing, cfg.world_seed))
KS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [8, 16, 32, 64]
for k in KS:
    for pair in (1, 0):
        for sym in (0, 1):
            bt.reset(); bt.set_update_mode(k, bool(sym)); bt.set_delayed_pairing(bool(pair))
            bt.run_known(0, 1 + W)
            st = bt.run_known(1 + W, 1 + W + K, time_kernels=True)
            print(f"B={B} k={k:2d} pairs={pair} symmetric={sym}: {st['corrections'] / (st['elapsed_ms'] * 1e-3):10.0f} corrections/s, "
                  f"{st['rank2_launches']} flushes of {st['rank2_ms'] / max(st['rank2_launches'], 1):6.2f} ms, gain steps "
                  f"{(st['elapsed_ms'] - st['rank2_ms']) / (2 * K) * 1e3:7.1f} us per correction", flush=True)
bt.close()
