"""Parity metric shared by the tests (SURVEY.md section 7 'Hard parts'): the covariance spans a 1e6
dynamic range (landmark variances 100, pose block 1e-4), so errors are taken PER BLOCK -- pose 3x3,
cross 3x2n and 2nx3, map 2nx2n -- each relative to that block's own max-abs; the state is split
theta | x,y | landmarks the same way (absolute when the reference block is ~0)."""
import numpy as np

FP64_TOL = 1e-9  # BASELINE.json north_star: state/covariance within 1e-9 relative of the CPU reference


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    if not (np.all(np.isfinite(a)) and np.all(np.isfinite(b))):
        return float("inf")
    scale = max(float(np.abs(b).max()), 1.0e-3)
    return float(np.abs(a - b).max() / scale)


def state_err(s, ref):
    return {"theta": _rel(s[0:1], ref[0:1]), "xy": _rel(s[1:3], ref[1:3]), "map": _rel(s[3:], ref[3:])}


def cov_err(c, ref):
    return {"pose": _rel(c[:3, :3], ref[:3, :3]), "cross_r": _rel(c[:3, 3:], ref[:3, 3:]),
            "cross_c": _rel(c[3:, :3], ref[3:, :3]), "map": _rel(c[3:, 3:], ref[3:, 3:])}


def worst(s, c, sref, cref):
    e = {}
    e.update({"state_" + k: v for k, v in state_err(s, sref).items()})
    e.update({"cov_" + k: v for k, v in cov_err(c, cref).items()})
    return max(e.values()), e


def assert_parity(s, c, sref, cref, tol=FP64_TOL, what=""):
    w, e = worst(s, c, sref, cref)
    assert np.isfinite(w) and w <= tol, f"{what}: per-block relative error {e} exceeds {tol}"
    return w
