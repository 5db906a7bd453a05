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


# ---- the path the published N = 10003 figure comes from: the main kernel runs whole rounds of 256 x 128 tiles (grouped
# ---- order, XCD-remapped), k_gemm_f32_tail finishes the rest of the list and the bottom strip of an ld that is an odd
# ---- multiple of 128 as 64 x 64 quarters behind it (ekf_dense.hip launch_dense_gemm).  Which kernel computes
# ---- which 128 x 128 block comes from the library itself (ekf_dense_tile_map: the host side of the kernels' own maps).

def test_dense_tail_path_exact_and_fp64(hip):
    """N = 4700: ld = 4736 = 37 x 128 -> 18 x 37 = 666 tiles of 256 x 128 = 512 (one full round, three
    tile groups incl. a ragged one) + 154 left over = 308 tail tiles of 128 x 128, + the bottom strip of 37: 345 tail tiles.
    (1) exact-integer operands, all three matrices asymmetric, + Q: every element must be bit-exact, which pins the
    tile -> (row, col) maps of BOTH kernels; (2) random dense F against fp64 per block."""
    N = 4700
    d = hip.DensePropagator(N)
    info = d.launch_info()
    assert info["ld"] == 4736 and info["tiles"] == 37
    assert info["n_big"] == 512 and info["n_tail"] == 345, info   # the test cannot silently take the one-kernel path
    mask = d.tile_map()
    assert mask.sum() == 345 and mask[36].all() and not mask[0, :8].any()

    rng = np.random.default_rng(29)
    B = rng.integers(-1, 2, size=(N, N)).astype(np.float32)   # {-1, 0, 1}: every partial sum stays far below 2^24
    Cm = rng.integers(-1, 2, size=(N, N)).astype(np.float32)
    Q = rng.integers(-5, 6, size=(N, N)).astype(np.float32)
    want = B.astype(np.float64) @ Cm.astype(np.float64) @ B.T.astype(np.float64) + Q
    assert np.abs(want).max() < 2 ** 23
    d.set(B, Cm, Q)
    d.propagate(1)
    got = d.sigma
    bad = got.astype(np.float64) != want
    if bad.any():
        r, c = np.argwhere(bad)[0]
        raise AssertionError(f"{bad.sum()} wrong elements, first at ({r},{c}) tile ({r // 128},{c // 128}) "
                             f"tail={mask[r // 128, c // 128]}: got {got[r, c]} want {want[r, c]}")
    # both products ran (T = B C is NN, T B^T is NT); a second application feeds the result back
    d.set(np.eye(N, dtype=np.float32), got, np.zeros((N, N), dtype=np.float32))
    d.propagate(1)
    assert np.array_equal(d.sigma, got)

    F = (np.eye(N) + rng.normal(size=(N, N)) / np.sqrt(N)).astype(np.float32)
    A = rng.normal(size=(N, 96))
    S = (A @ A.T / 96 + np.eye(N)).astype(np.float32)
    Qd = np.diag(rng.uniform(1e-4, 1e-2, size=N)).astype(np.float32)
    d.set(F, S, Qd)
    d.propagate(1)
    got = d.sigma.astype(np.float64)
    want = _ref(F, S, Qd)
    assert max(cov_err(got, want).values()) < FP32_TOL
    # ... and tile by tile, so a wrong tail tile cannot hide in a per-block maximum taken over the whole matrix
    scale = np.abs(want).max()
    for tm, tn in np.argwhere(mask):
        r0, c0 = tm * 128, tn * 128
        blk_g, blk_w = got[r0:r0 + 128, c0:c0 + 128], want[r0:r0 + 128, c0:c0 + 128]
        assert np.abs(blk_g - blk_w).max() / scale < FP32_TOL, (tm, tn)
    d.close()


def test_dense_full_size_n5000_rows_in_tail_tiles(hip):
    """BASELINE.json configs[3] at its full size N = 10003 (n = 5000, ld = 10112 = 79 x 128: 39 x 79 = 3081 tiles of
    256 x 128 = 6 rounds of 512 + 9 left over -> 18 tail tiles, + the bottom strip of 79: 97 tail tiles of 128 x 128): rows
    sampled INSIDE tail tiles (and a few outside) against fp64."""
    N = 10003
    d = hip.DensePropagator(N)
    info = d.launch_info()
    assert info["ld"] == 10112 and info["tiles"] == 79 and info["n_big"] == 3072 and info["n_tail"] == 97, info
    mask = d.tile_map()
    assert mask.sum() == 97 and mask[78].all()
    rng = np.random.default_rng(4)
    F = np.eye(N, dtype=np.float32) + rng.standard_normal((N, N), dtype=np.float32) * np.float32(0.05 / np.sqrt(N))
    A = rng.standard_normal((N, 64), dtype=np.float32)
    S = A @ A.T / np.float32(64) + np.eye(N, dtype=np.float32)
    Q = np.zeros((N, N), dtype=np.float32)
    Q[0, 0] = Q[1, 1] = Q[2, 2] = 1e-4
    d.set(F, S, Q)
    d.propagate(1)
    got = d.sigma
    tail_rows_blocks = np.unique(np.argwhere(mask)[:, 0])
    rows = []
    for tm in tail_rows_blocks:                       # one row in every block row that holds tail tiles
        rows.append(min(N - 1, int(tm) * 128 + int(rng.integers(0, 128))))
    rows += [0, 2, 5000, int(rng.integers(0, 9000))]  # and outside
    rows = np.array(sorted(set(rows)))
    F64 = F.astype(np.float64)
    want = (F64[rows] @ S.astype(np.float64)) @ F64.T + Q[rows].astype(np.float64)
    scale = np.abs(want).max()
    err = np.abs(got[rows].astype(np.float64) - want) / scale
    assert err.max() < FP32_TOL, f"rel err {err.max():.2e}"
    # the sampled rows really cross tail tiles: check those column ranges on their own
    hit = 0
    for k, r in enumerate(rows):
        for tn in np.nonzero(mask[r // 128])[0]:
            c0 = int(tn) * 128
            assert err[k, c0:min(N, c0 + 128)].max() < FP32_TOL
            hit += 1
    assert hit >= len(tail_rows_blocks)
    d.close()
