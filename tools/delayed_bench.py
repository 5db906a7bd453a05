"""Delayed rank-2k update throughput on the GPU box (configs[4] share)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K, W = 32, 4
log = synth.make_known_log(synth.config5(filters=B, steps=1 + W + K, n=1000))
bt = capi.BatchEKF(B, 1000)
bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
for k in (0, 4, 8, 16, 32, 64):
    bt.reset(); bt.set_update_mode(k)
    bt.run_known(0, 1 + W)
    st = bt.run_known(1 + W, 1 + W + K, time_kernels=True)
    print(f"B={B} k={k:2d}: {st['corrections'] / (st['elapsed_ms'] * 1e-3):10.0f} corrections/s, elapsed {st['elapsed_ms']:8.2f} ms, "
          f"covariance passes {st['rank2_launches']} taking {st['rank2_ms']:.2f} ms "
          f"({st['rank2_bytes_per_launch'] * st['rank2_launches'] / (st['rank2_ms'] * 1e-3) / 1e9 if st['rank2_ms'] else 0:.0f} GB/s), other {st['elapsed_ms'] - st['rank2_ms']:.2f} ms", flush=True)
print(bt.checksum())
