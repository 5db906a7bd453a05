"""GPU box: the rank-2 stream of a pool at several map sizes: plain kernel vs row-packed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth

for n, B in ((60, 65535), (100, 65535), (200, 16384), (200, 65535), (333, 8192), (333, 24000), (500, 4096), (1000, 1024)):
    B = min(B, 65535)
    cfg = synth.config5(filters=B, steps=6, n=n)
    world = synth.make_world(n, cfg.half_extent, cfg.min_spacing, cfg.world_seed)
    bt = capi.BatchEKF(B, n)
    bt.simulate_known_log(cfg, world)
    bt.run_known(0, 2)
    out = []
    for rows, label, u in ((-1, "plain", 0), (0, "auto", 0), (8, "v8", 0), (24, "v24", 0), (32, "v32", 0)):
        bt.set_tuning(rows, -1, u)
        st = bt.run_known(2, 6, time_kernels=True)
        out.append(st["rank2_bytes_per_launch"] / (st["rank2_ms"] / st["rank2_launches"] * 1e-3) / 1e9)
    N = 3 + 2 * n
    print(f"n={n:5d} B={B:6d} (pool {B * N * ((N + 15) // 16 * 16) * 8 / 1e9:6.1f} GB): plain {out[0]:7.0f} GB/s ({out[0] / 8000:.3f}) | "
          f"packed: auto {out[1]:7.0f} ({out[1] / 8000:.3f}) vrows8 {out[2]:7.0f} vrows24 {out[3]:7.0f} vrows32 {out[4]:7.0f}", flush=True)
    bt.close()
