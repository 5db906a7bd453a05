"""-m gpu: dense general-F covariance propagation Sigma <- F Sigma F^T + Q in fp32 on the matrix cores
(BASELINE.json configs[3]) against fp64 references.  Tolerance 1e-4 relative per block (SURVEY.md
section 8(d): fp32 compute with fp64 check)."""
import numpy as np
import pytest

from parity import cov_err

pytestmark = pytest.mark.gpu
FP32_TOL = 1e-4


def _ref(F, S, Q, iters=1):
    F, S, Q = F.astype(np.float64), S.astype(np.float64), Q.astype(np.float64)
    for _ in range(iters):
        S = F @ S @ F.T + Q
    return S


@pytest.mark.parametrize("N", [43, 128, 300, 403])
def test_dense_random_F_vs_fp64(hip, N):
    rng = np.random.default_rng(N)
    F = (np.eye(N) + rng.normal(size=(N, N)) / np.sqrt(N)).astype(np.float32)   # dense, asymmetric
    A = rng.normal(size=(N, N))
    S = (A @ A.T / N + np.eye(N)).astype(np.float32)                            # SPD covariance
    Q = np.diag(rng.uniform(1e-4, 1e-2, size=N)).astype(np.float32)
    d = hip.DensePropagator(N)
    d.set(F, S, Q)
    d.propagate(1)
    got = d.sigma.astype(np.float64)
    want = _ref(F, S, Q)
    assert max(cov_err(got, want).values()) < FP32_TOL
    d.propagate(2)  # result feeds back as the next Sigma
    assert max(cov_err(d.sigma.astype(np.float64), _ref(F, want.astype(np.float32), Q, 2)).values()) < 5 * FP32_TOL
    d.close()


def test_dense_operand_layouts_exact(hip):
    """Integer data makes every product exact in fp32: catches a transposed operand or a swapped C/D map
    (A = I with an ASYMMETRIC B, as the MFMA guide prescribes)."""
    N = 200
    rng = np.random.default_rng(0)
    B = rng.integers(-3, 4, size=(N, N)).astype(np.float32)  # asymmetric
    I = np.eye(N, dtype=np.float32)
    Z = np.zeros((N, N), dtype=np.float32)
    d = hip.DensePropagator(N)
    d.set(I, B, Z); d.propagate(1)
    assert np.array_equal(d.sigma, B)                       # I B I^T
    d.set(B, I, Z); d.propagate(1)
    assert np.array_equal(d.sigma, B @ B.T)                 # B I B^T
    C = rng.integers(-2, 3, size=(N, N)).astype(np.float32)
    Q = rng.integers(-5, 6, size=(N, N)).astype(np.float32)
    d.set(B, C, Q); d.propagate(1)
    assert np.array_equal(d.sigma, B @ C @ B.T + Q)         # all three asymmetric
    d.close()


def test_dense_reproduces_the_structured_prediction(hip, oracle):
    """F = At = I + A of the reference's motion model (ekf_slam.cpp:85-101): the dense fp32 path must agree
    with the fp64 prediction() to fp32 accuracy."""
    n = 200
    o = oracle.OracleEKF(n, oracle.STRUCTURED)
    rng = np.random.default_rng(7)
    N = o.N
    A = rng.normal(size=(N, N))
    S0 = A @ A.T / N + np.eye(N)
    st = np.zeros(N); st[0] = 0.3
    o.state, o.cov = st, S0
    dth, dx = 0.05, 0.02
    o.prediction(dth, dx)
    th = 0.3
    At = np.eye(N)
    At[1, 0] += -(dx / dth) * np.cos(th) + (dx / dth) * np.cos(th + dth)
    At[2, 0] += -(dx / dth) * np.sin(th) + (dx / dth) * np.sin(th + dth)
    Q = np.zeros((N, N)); Q[0, 0] = Q[1, 1] = Q[2, 2] = 1e-4
    d = hip.DensePropagator(N)
    d.set(At.astype(np.float32), S0.astype(np.float32), Q.astype(np.float32))
    d.propagate(1)
    assert max(cov_err(d.sigma.astype(np.float64), o.cov).values()) < FP32_TOL
    d.close()
