import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekf_slam_ml_amd import capi, synth
B, K, W = 4096, 16, 4
cfg = synth.config5(filters=B, steps=1 + W + K, n=1000)
bt = capi.BatchEKF(B, 1000)
bt.simulate_known_log(cfg, synth.make_world(1000, cfg.half_extent, cfg.min_spacing, cfg.world_seed))
bt.set_active_set(True)
bt.run_known(0, 1 + W)
st = bt.run_known(1 + W, 1 + W + K, time_kernels=True)
print(st, st["corrections"] / st["elapsed_ms"] * 1e3)
