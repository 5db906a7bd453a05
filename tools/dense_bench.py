"""configs[3] on the GPU box: dense F Sigma F^T + Q, fp32 MFMA, N = 10003 (n = 5000)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10003
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(4)
t0 = time.time()
F = (np.eye(N, dtype=np.float32) + (rng.standard_normal((N, N), dtype=np.float32) / np.float32(np.sqrt(N)) * np.float32(0.05)))
A = rng.standard_normal((N, 64), dtype=np.float32)
S = (A @ A.T / np.float32(64) + np.eye(N, dtype=np.float32))
Q = np.zeros((N, N), dtype=np.float32); Q[0, 0] = Q[1, 1] = Q[2, 2] = 1e-4
print("host gen", time.time() - t0, flush=True)
d = capi.DensePropagator(N)
d.set(F, S, Q)
d.propagate(1)
got = d.sigma
# fp64 check on sampled rows (full fp64 N^3 on the host would take minutes)
rows = rng.choice(N, size=8, replace=False)
F64, S64 = F.astype(np.float64), S.astype(np.float64)
want = (F64[rows] @ S64) @ F64.T + Q[rows].astype(np.float64)
err = np.abs(got[rows] - want).max() / np.abs(want).max()
print(f"fp64 spot check (8 rows): rel err {err:.2e}", flush=True)
d.set(F, S, Q)
ms = [d.propagate(1) for _ in range(iters)]
flop = 4.0 * N ** 3
ld = (N + 127) // 128 * 128
print(f"N={N} ld={ld}: median {np.median(ms):.2f} ms per propagation, {flop / (np.median(ms) * 1e-3) / 1e12:.1f} TFLOP/s algorithmic "
      f"({4.0 * ld ** 3 / (np.median(ms) * 1e-3) / 1e12:.1f} incl. padding), {1e3 / np.median(ms):.1f} predicts/s; min {min(ms):.2f} ms")
