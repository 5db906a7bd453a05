"""-m gpu: seeded differential fuzzing of the drop-in API against the dense-literal checker: random map
sizes, random visibility patterns, both prediction() branches, angle wrap-around, measurement() and
data_association() interleaved on the same object, random update modes -- every scenario must stay
within the north_star tolerance and make identical association decisions."""
import numpy as np
import pytest

from parity import FP64_TOL, assert_parity

pytestmark = pytest.mark.gpu


def _scenario(hip, oracle, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 91)) if seed % 4 else int(rng.integers(52, 140))   # every 4th scenario beyond the small-map path
    f, o = hip.EKF_SLAM(n), oracle.OracleEKF(n, oracle.DENSE)
    mode = int(rng.choice([0, 0, 3, 16]))
    f.set_update_mode(mode, symmetric_gather=False)
    f.set_small_map_path(bool(rng.integers(0, 2)))
    f.set_active_prefix(bool(rng.integers(0, 2)))
    rng.integers(0, 2)   # (draw of a form that left the library in round 4: the scenarios keep their inputs)
    rng2 = np.random.default_rng(seed + 7919)   # (a second stream: the scenarios of round 1 keep their inputs)
    f.set_call_fused(bool(rng2.integers(0, 2)))                                   # two launches per measurement() call
    rng2.integers(0, 2), rng2.choice([0, 3, 17, 64])   # (draws of a path that left the library: the scenarios keep their inputs)
    world = rng.uniform(-2.5, 2.5, size=(n, 2))
    world[np.hypot(world[:, 0], world[:, 1]) < 0.3] += 0.6          # keep landmarks off the start pose
    pose = np.zeros(3)                                              # true (theta, x, y)
    known_f, known_o = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    assoc_p = float(rng.choice([0.0, 0.0, 0.5, 1.0]))   # share of steps that go through data_association()
    for t in range(int(rng.integers(6, 22))):
        if rng.random() < 0.1:                          # live switches must not disturb the filter
            rng.integers(0, 2)   # (see above)
        if rng2.random() < 0.1:
            f.set_call_fused(bool(rng2.integers(0, 2)))
        if rng2.random() < 0.1:
            rng2.integers(0, 2), rng2.choice([0, 3, 17, 64])
        if rng.random() < 0.05:
            mode = int(rng.choice([0, 3, 16]))
            f.set_update_mode(mode, symmetric_gather=False)
        use_assoc = rng.random() < assoc_p
        dth = float(rng.choice([0.0, 5e-7, rng.normal(0, 0.3), rng.normal(0, 1.5)]))   # both branches of :79
        dx = float(rng.normal(0.05, 0.05))
        pose[1] += dx * np.cos(pose[0]); pose[2] += dx * np.sin(pose[0]); pose[0] += dth
        f.prediction((dth, dx)); o.prediction(dth, dx)
        c, s = np.cos(pose[0]), np.sin(pose[0])
        d = world - pose[1:]
        rf = np.stack([c * d[:, 0] + s * d[:, 1], -s * d[:, 0] + c * d[:, 1]], axis=1) + rng.normal(0, 0.004, size=(n, 2))
        if use_assoc and t > 0:
            k = int(rng.integers(0, min(n, 6) + 1))
            pick = rng.choice(n, size=k, replace=False)
            a, b = f.data_association(rf[pick], known_f), o.data_association(rf[pick], known_o)
            assert np.array_equal(a, b), f"seed {seed} step {t}: {a} vs {b}"
            assert np.array_equal(known_f, known_o)
        else:
            vis = (rng.random(n) < rng.choice([0.0, 0.2, 0.7, 1.0])).astype(np.uint8) if t else np.zeros(n, dtype=np.uint8)
            f.measurement(rf.reshape(-1), vis); o.measurement(rf.reshape(-1), vis)
        if rng.random() < 0.15:
            assert_parity(f.state, f.cov, o.state, o.cov, FP64_TOL, f"seed {seed} step {t} (n={n}, mode={mode})")
    assert_parity(f.state, f.cov, o.state, o.cov, FP64_TOL, f"seed {seed} final (n={n}, mode={mode})")
    f.close()


@pytest.mark.parametrize("block", range(6))
def test_differential_fuzz(hip, oracle, block):
    for seed in range(block * 10, block * 10 + 10):
        _scenario(hip, oracle, 1000 + seed)
