"""Tuning sweep of the rank-2 covariance kernel on the GPU box (interleaved rounds, one process)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
log = synth.make_known_log(synth.config5(filters=B, steps=4, n=n))
bf = capi.BatchEKF(B, n)
bf.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
bf.run_known(0, 2)
variants = [(rows, 1, u) for u, rr in ((4, (4, 8)), (8, (8, 16, 24, 32)), (16, (16, 32, 48, 64))) for rows in rr]
res = {v: [] for v in variants}
for rd in range(rounds):
    for v in variants:
        bf.set_tuning(*v)
        st = bf.run_known(2, 4, time_kernels=True)
        res[v].append(st["rank2_bytes_per_launch"] / (st["rank2_ms"] / st["rank2_launches"] * 1e-3) / 1e9)
out = sorted(((np.median(x), min(x), max(x), v) for v, x in res.items()), reverse=True)
for med, lo, hi, v in out:
    print(f"rows={v[0]:3d} nt={v[1]} U={v[2]}: median {med:7.0f} GB/s  [{lo:.0f}, {hi:.0f}]", flush=True)
