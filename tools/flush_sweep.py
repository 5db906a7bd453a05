import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K, W = 32, 2
cfg = synth.config5(filters=B, steps=1 + W + K, n=1000)
bt = capi.BatchEKF(B, 1000)
bt.simulate_known_log(cfg, synth.make_world(1000, cfg.half_extent, cfg.min_spacing, cfg.world_seed))
for k in (16, 32):
    for sym in (0, 1):
        rows = 16
        bt.reset(); bt.set_update_mode(k, sym); bt.set_tuning(rows, -1, 0)
        bt.run_known(0, 1 + W)
        st = bt.run_known(1 + W, 1 + W + K, time_kernels=True)
        print(f"k={k} sym={sym}: {st['corrections'] / (st['elapsed_ms'] * 1e-3):10.0f} corr/s, flush {st['rank2_ms'] / st['rank2_launches']:.2f} ms each x{st['rank2_launches']}, "
              f"other {st['elapsed_ms'] - st['rank2_ms']:.2f} ms", flush=True)
