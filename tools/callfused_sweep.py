"""GPU box: streaming pass of the call-fused update (k_rank2v) at B = 4096 (or argv[1]), n = 1000, V = 2: tuning sweep."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 1000
cfg = synth.config5(filters=B, steps=12, n=n)
world = synth.make_world(n, cfg.half_extent, cfg.min_spacing, cfg.world_seed)
bt = capi.BatchEKF(B, n)
bt.simulate_known_log(cfg, world)
for cf in (False, True):
    for rows, u in ((0, 0), (32, 8), (64, 16), (64, 8), (16, 8), (128, 16)):
        bt.reset(); bt.set_call_fused(cf); bt.set_tuning(rows, -1, u)
        bt.run_known(0, 4)
        st = bt.run_known(4, 12, time_kernels=True)
        ms = st["rank2_ms"] / st["rank2_launches"]
        print(f"call_fused={cf} rows={rows:3d} group={u:2d}: pass {ms:7.3f} ms = {st['rank2_bytes_per_launch'] / ms / 1e6:7.1f} GB/s "
              f"({st['rank2_bytes_per_launch'] / ms / 1e6 / 8000:.3f} of 8 TB/s), {st['corrections'] / st['elapsed_ms'] * 1e3:9.0f} update steps/s", flush=True)
bt.close()
