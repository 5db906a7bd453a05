"""-m gpu: batched unknown data association (ekf_batch_upload_unknown_log / ekf_batch_run_unknown) -- the node
loop of nuslam/src/unknown_data_assoc.cpp:300-323 for B independent robots -- against the CPU checker run
filter by filter: decisions and known counts identical, state/covariance within FP64_TOL."""
import numpy as np
import pytest

from ekf_slam_ml_amd import synth
from parity import FP64_TOL, assert_parity

pytestmark = pytest.mark.gpu


def _oracle_replay(oracle, log, b, n, t0, t1, o=None, known=None):
    if o is None:
        o, known = oracle.OracleEKF(n, oracle.DENSE), np.zeros(n, dtype=np.uint8)
    dec = np.full((t1 - t0, log.meas_xy.shape[2]), -2, dtype=np.int32)
    for t in range(t0, t1):
        J = int(log.count[t, b])
        o.prediction(*log.twist[t, b])
        dec[t - t0, :J] = o.data_association(log.meas_xy[t, b, :J], known)
    return o, known, dec


def _ragged_log(n, B, T, seed, vmax):
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=seed, half_extent=1.5, min_spacing=0.25,
                          max_visible_dis=0.7, vmax=vmax)
    return synth.make_unknown_log(cfg)


def test_batch_unknown_vs_oracle(hip, oracle):
    n, B, T = 20, 6, 60
    log = _ragged_log(n, B, T, 777, 6)
    assert len(set(log.count.reshape(-1).tolist())) > 2, "the slots must be ragged across filters"
    snap = []
    for small in (False, True):  # four launches per measurement slot / one LDS-resident launch per step (N = 43)
        bt = hip.BatchEKF(B, n)
        bt.set_small_map_path(small)
        bt.set_step_fused(False)
        bt.upload_unknown_log(log.twist, log.count, log.meas_xy)
        st = bt.run_unknown(0, T, time_kernels=True)
        dec, kc = bt.decisions(), bt.known_counts()
        applied = 0
        for b in range(B):
            o, known, d = _oracle_replay(oracle, log, b, n, 0, T)
            assert np.array_equal(dec[:, b], d), f"filter {b}: decisions differ"
            assert kc[b] == int(known.sum()) and known[:kc[b]].all()
            assert_parity(bt.state(b), bt.cov(b), o.state, o.cov, FP64_TOL, f"batch unknown, filter {b}")
            applied += int((d >= 0).sum())
        assert st["corrections"] == applied and applied > 100
        launches = int((log.count.max(axis=1) > 0).sum()) if small else int(log.count.max(axis=1).sum())
        assert st["filter_steps"] == B * T and st["rank2_launches"] == launches
        snap.append((dec.copy(), [bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)]))
        bt.close()
    assert np.array_equal(snap[0][0], snap[1][0])
    for b in range(B):  # same arithmetic in the same order: bit-identical
        assert np.array_equal(snap[0][1][b], snap[1][1][b]) and np.array_equal(snap[0][2][b], snap[1][2][b])


def test_batch_unknown_split_runs_and_prefix_off(hip, oracle):
    """Two successive runs continue from the device-resident known counts; the discovered-prefix confinement is
    exact, so turning it off must not change a single bit."""
    n, B, T = 30, 4, 40
    log = _ragged_log(n, B, T, 31337, 5)
    out = []
    for prefix in (1, 0):
        bt = hip.BatchEKF(B, n)
        bt.set_active_prefix(bool(prefix))
        bt.upload_unknown_log(log.twist, log.count, log.meas_xy)
        if prefix:
            bt.run_unknown(0, 17)
            bt.run_unknown(17, T)
        else:
            bt.run_unknown(0, T)
        out.append((bt.decisions().copy(), bt.known_counts().copy(), [bt.state(b) for b in range(B)],
                    [bt.cov(b) for b in range(B)]))
        bt.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    for b in range(B):
        assert np.array_equal(out[0][2][b], out[1][2][b]) and np.array_equal(out[0][3][b], out[1][3][b])
    o, known, d = _oracle_replay(oracle, log, 2, n, 0, T)
    assert np.array_equal(out[0][0][:, 2], d)
    assert_parity(out[0][2][2], out[0][3][2], o.state, o.cov, FP64_TOL, "split runs")


def test_batch_unknown_matches_single_filter_api(hip):
    """The batch path and ekf_associate share kernels: the same log through both is bit-identical."""
    n, B, T = 40, 3, 25
    log = _ragged_log(n, B, T, 99, 7)
    bt = hip.BatchEKF(B, n)
    bt.upload_unknown_log(log.twist, log.count, log.meas_xy)
    bt.run_unknown()
    for b in range(B):
        f = hip.EKF_SLAM(n)
        f.set_small_map_path(False)
        k = np.zeros(n, dtype=np.uint8)
        for t in range(T):
            f.prediction(log.twist[t, b])
            f.data_association(log.meas_xy[t, b, :log.count[t, b]], k)
        assert np.array_equal(f.state, bt.state(b)) and np.array_equal(f.cov, bt.cov(b))
        assert int(k.sum()) == bt.known_counts()[b]
        f.close()
    bt.close()


def test_batch_unknown_map_full_and_empty_steps(hip, oracle):
    """n = 2 with three distinct tubes in view: the third is dropped for good (ekf_slam.cpp:294,330 -- idx == n);
    steps without measurements only predict."""
    n, B, T = 2, 2, 6
    twist = np.zeros((T, B, 2)); twist[:, :, 1] = 0.01
    count = np.zeros((T, B), dtype=np.int32)
    meas = np.zeros((T, B, 3, 2))
    tubes = np.array([[0.5, 0.0], [0.0, 0.5], [-0.5, 0.0]])
    for t in (1, 3, 4):
        count[t, 0] = 3
        meas[t, 0] = tubes - np.array([0.01 * (t + 1), 0.0])
    count[2, 1] = 1; meas[2, 1, 0] = [0.3, 0.3]
    log = type("L", (), {"twist": twist, "count": count, "meas_xy": meas})
    bt = hip.BatchEKF(B, n)
    bt.upload_unknown_log(twist, count, meas)
    bt.run_unknown()
    dec = bt.decisions()
    for b in range(B):
        o, known, d = _oracle_replay(oracle, log, b, n, 0, T)
        assert np.array_equal(dec[:, b], d)
        assert_parity(bt.state(b), bt.cov(b), o.state, o.cov, FP64_TOL, f"map full, filter {b}")
    assert (dec[[1, 3, 4], 0, 2] == -1).all() and bt.known_counts().tolist() == [2, 1]
    bt.close()


def test_batch_unknown_errors(hip):
    bt = hip.BatchEKF(2, 5)
    with pytest.raises(hip.EkfError):
        bt.uT = 1
        bt.run_unknown(0, 1)  # nothing uploaded
    with pytest.raises(hip.EkfError):
        bt.upload_unknown_log(np.zeros((1, 2, 2)), np.full((1, 2), 4, dtype=np.int32), np.zeros((1, 2, 3, 2)))  # count > jmax
    bt.upload_unknown_log(np.zeros((2, 2, 2)), np.zeros((2, 2), dtype=np.int32), np.zeros((2, 2, 3, 2)))
    with pytest.raises(hip.EkfError):
        bt.run_unknown(0, 3)
    st = bt.run_unknown()
    assert st["corrections"] == 0 and st["rank2_launches"] == 0
    bt.close()


def test_batch_unknown_outgrows_the_small_path(hip, oracle):
    """Three new landmarks per step: the discovered prefix passes N = 104 mid-run, so the run switches from one
    LDS-resident launch per step to four launches per slot -- and must not differ from the all-multi-launch run."""
    n, B, T, J = 90, 2, 34, 5
    rng = np.random.default_rng(12)
    grid = np.array([[0.6 * (k % 12) - 3.27, 0.6 * (k // 12) - 2.03] for k in range(120)])
    twist = np.zeros((T, B, 2)); twist[:, :, 1] = 0.002; twist[:, 1, 0] = 0.001
    count = np.zeros((T, B), dtype=np.int32)
    meas = np.zeros((T, B, J, 2))
    for t in range(T):
        new = [3 * t, 3 * t + 1, 3 * t + 2]
        old = [t, (2 * t) // 3] if t > 0 else []
        ids = new + old
        count[t, 0] = len(ids)
        meas[t, 0, :len(ids)] = grid[ids] + rng.normal(0, 0.003, (len(ids), 2))
        ids1 = (old + new)[:4]          # ragged: other order, one reading fewer
        count[t, 1] = len(ids1)
        meas[t, 1, :len(ids1)] = grid[ids1] + rng.normal(0, 0.003, (len(ids1), 2))
    log = type("L", (), {"twist": twist, "count": count, "meas_xy": meas})
    snap = []
    for small, fused in ((True, False), (False, False), (True, True)):
        bt = hip.BatchEKF(B, n)
        bt.set_small_map_path(small)
        bt.set_step_fused(fused)
        bt.upload_unknown_log(twist, count, meas)
        st = bt.run_unknown(0, 20)
        st2 = bt.run_unknown(20, T)
        launches = st["rank2_launches"] + st2["rank2_launches"]
        if fused:    # LDS-resident step kernel first, then the any-size step kernel: one launch per step throughout
            assert launches == int((count.max(axis=1) > 0).sum())
        elif small:  # early steps took the one-launch form, late ones (known_count + readings > 50) could not
            assert T < launches < int(count.max(axis=1).sum())
        else:
            assert launches == int(count.max(axis=1).sum())
        snap.append((bt.decisions().copy(), bt.known_counts().copy(), [bt.state(b) for b in range(B)],
                     [bt.cov(b) for b in range(B)]))
        bt.close()
    assert snap[0][1].max() > 52  # N_b > 104: beyond the small path
    for other in snap[1:]:
        assert np.array_equal(snap[0][0], other[0]) and np.array_equal(snap[0][1], other[1])
        for b in range(B):
            assert np.array_equal(snap[0][2][b], other[2][b]) and np.array_equal(snap[0][3][b], other[3][b])
    o, known, d = _oracle_replay(oracle, log, 0, n, 0, T)
    assert np.array_equal(snap[0][0][:, 0], d)
    assert_parity(snap[0][2][0], snap[0][3][0], o.state, o.cov, FP64_TOL, "outgrowing the small path")


@pytest.mark.parametrize("n", [30, 70])   # LDS-resident paths / multi-kernel paths
def test_batch_unknown_then_known_then_unknown(hip, oracle, n):
    """The two node loops on the same pool: after a known-association run (which corrects arbitrary indices and, on its
    first call, re-initialises every landmark, ekf_slam.cpp:113-128) the discovered-prefix shortcut must be off, and the
    device-resident known counts must still carry over."""
    B, T = 3, 36
    ulog = _ragged_log(n, B, T, 2468, 5)
    kcfg = synth.SimConfig(n=n, steps=T, filters=B, seed=2468, half_extent=1.5, min_spacing=0.25, max_visible_dis=0.7, vmax=6)
    klog = synth.make_known_log(kcfg)
    klog.lm_idx[12] = -1   # step 12 is this pool's FIRST measurement() call: landmark initialisation only
    bt = hip.BatchEKF(B, n)
    bt.upload_unknown_log(ulog.twist, ulog.count, ulog.meas_xy)
    bt.upload_known_log(klog.twist, klog.lm_idx, klog.z_xy, klog.init_xy)
    bt.run_unknown(0, 12)
    bt.run_known(12, 24)
    bt.run_unknown(24, T)
    dec = bt.decisions()
    for b in range(B):
        o, known = oracle.OracleEKF(n, oracle.DENSE), np.zeros(n, dtype=np.uint8)
        for t in range(T):
            if 12 <= t < 24:
                o.prediction(*klog.twist[t, b])
                sensor, vis = klog.expand_step(t, b)
                if t == 12:   # the first measurement() call of this object: the log's init vector, nothing visible
                    sensor, vis = klog.init_xy[b].copy(), np.zeros(n, dtype=np.uint8)
                o.measurement(sensor, vis)
            else:
                J = int(ulog.count[t, b])
                o.prediction(*ulog.twist[t, b])
                a = o.data_association(ulog.meas_xy[t, b, :J], known)
                assert np.array_equal(dec[t, b, :J], a), f"filter {b} step {t}"
        assert bt.known_counts()[b] == int(known.sum())
        assert_parity(bt.state(b), bt.cov(b), o.state, o.cov, FP64_TOL, f"mixed loops, filter {b}")
    bt.close()


def test_surveyed_map_then_unknown_association(hip, oracle):
    """ekf_batch_set_known_counts: a map built through the known-association path (whose first call initialises all n
    landmarks, ekf_slam.cpp:113-128) is then used with unknown association, known_list all true from the start -- the
    batch twin of passing a prefilled known_list to data_association().  Every reading is scored against ALL n
    landmarks and corrected at full width; decisions, state and covariance against the dense checker."""
    n, B, T = 40, 3, 25
    ucfg = synth.SimConfig(n=n, steps=T, filters=B, seed=99, half_extent=2.0, min_spacing=0.35, max_visible_dis=0.9, vmax=5)
    ulog = synth.make_unknown_log(ucfg)
    world = ulog.world
    rng = np.random.default_rng(5)
    vm = 16
    Ta = 1 + (n + vm - 1) // vm
    lm = np.full((Ta, B, vm), -1, dtype=np.int32)
    z = np.zeros((Ta, B, vm, 2))
    for t in range(1, Ta):
        idx = np.arange((t - 1) * vm, min(n, t * vm))
        lm[t, :, :len(idx)] = idx
        z[t, :, :len(idx)] = world[idx][None] + rng.normal(0.0, 0.005, size=(B, len(idx), 2))
    init = (world[None] + rng.normal(0.0, 0.005, size=(B, n, 2))).reshape(B, 2 * n)
    tw0 = np.zeros((Ta, B, 2))
    snap = []
    for small in (True, False):   # LDS-resident step kernel (N = 83) / four launches per measurement slot
        bt = hip.BatchEKF(B, n)
        bt.set_small_map_path(small)
        bt.upload_known_log(tw0, lm, z, init)
        bt.run_known()
        bt.set_known_counts(n)
        assert np.array_equal(bt.known_counts(), np.full(B, n))
        bt.upload_unknown_log(ulog.twist, ulog.count, ulog.meas_xy)
        st = bt.run_unknown(0, T)
        dec = bt.decisions()
        matched = 0
        for b in range(B):
            o = oracle.OracleEKF(n, oracle.DENSE)
            for t in range(Ta):
                o.prediction(0.0, 0.0)
                o.measurement_compact(init[b], lm[t, b], z[t, b])
            known = np.ones(n, dtype=np.uint8)
            o, known, d = _oracle_replay(oracle, ulog, b, n, 0, T, o, known)
            assert np.array_equal(dec[:, b], d), f"filter {b}: decisions differ"
            assert_parity(bt.state(b), bt.cov(b), o.state, o.cov, FP64_TOL, f"surveyed map, filter {b}")
            good = d >= 0
            matched += int(good.sum())
            assert np.array_equal(d[good], ulog.truth_idx[:, b][good])   # and the association is the true one
        assert st["corrections"] == matched and matched > 50
        snap.append([(bt.state(b), bt.cov(b)) for b in range(B)])
        with pytest.raises(hip.EkfError):
            bt.set_known_counts(n + 1)
        bt.close()
    for (s0, c0), (s1, c1) in zip(*snap):
        assert np.array_equal(s0, s1) and np.array_equal(c0, c1)


@pytest.mark.parametrize("surveyed", [False, True])
def test_step_fused_beyond_the_small_path(hip, oracle, surveyed):
    """Prefixes beyond N_b = 104: one launch per step (ekf_stepfused.hip: scores / decision / gain against the stored
    covariance minus the step's pending pairs, ONE pass over the prefix at the end of the step) against four launches per
    measurement slot -- bit for bit -- and against the CPU checker (decisions identical, 1e-9)."""
    n, B, T = 150, 4, 30
    # a fast robot on a 1.7 m circle: new landmarks come into view every step, the discovered prefix passes 104 quickly
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=4242, half_extent=2.6, min_spacing=0.3, max_visible_dis=0.7, vmax=7,
                          v_cmd=2.0, w_cmd=1.2)
    log = synth.make_unknown_log(cfg)
    assert len(set(log.count.reshape(-1).tolist())) > 2
    world = log.world
    rng = np.random.default_rng(11)
    init = (world[None] + rng.normal(0.0, 0.005, size=(B, n, 2))).reshape(B, 2 * n)
    lm0 = np.full((2, B, 1), -1, dtype=np.int32)
    snap = []
    for fused in (True, False):
        bt = hip.BatchEKF(B, n)
        bt.set_small_map_path(False)   # (the LDS-resident step kernel would take the steps whose prefix is still <= 104)
        bt.set_step_fused(fused)
        if surveyed:   # every landmark known from the start (first known-association call initialises all of them)
            bt.upload_known_log(np.zeros((2, B, 2)), lm0, np.zeros((2, B, 1, 2)), init)
            bt.run_known()
            bt.set_known_counts(n)
        bt.upload_unknown_log(log.twist, log.count, log.meas_xy)
        bt.run_unknown(0, 11)
        st = bt.run_unknown(11, T, time_kernels=True)
        dec, kc = bt.decisions(), bt.known_counts()
        snap.append((dec.copy(), kc.copy(), [bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)], st))
        bt.close()
    assert np.array_equal(snap[0][0], snap[1][0]) and np.array_equal(snap[0][1], snap[1][1])
    for b in range(B):
        assert np.array_equal(snap[0][2][b], snap[1][2][b]), f"filter {b} state"
        assert np.array_equal(snap[0][3][b], snap[1][3][b]), f"filter {b} covariance"
    assert snap[0][4]["corrections"] == snap[1][4]["corrections"] > 50
    assert snap[0][4]["rank2_launches"] < snap[1][4]["rank2_launches"]   # one launch per step, not per slot
    for b in (0, B - 1):
        o = oracle.OracleEKF(n, oracle.STRUCTURED)
        known = np.zeros(n, dtype=np.uint8)
        if surveyed:
            o.prediction(0.0, 0.0); o.measurement_compact(init[b], lm0[0, b], np.zeros((1, 2)))
            o.prediction(0.0, 0.0); o.measurement_compact(init[b], lm0[1, b], np.zeros((1, 2)))
            known[:] = 1
        o, known, d = _oracle_replay(oracle, log, b, n, 0, T, o, known)
        assert np.array_equal(snap[0][0][:, b], d), f"filter {b}: decisions differ from the checker"
        assert_parity(snap[0][2][b], snap[0][3][b], o.state, o.cov, FP64_TOL, f"step-fused filter {b}")


@pytest.mark.parametrize("surveyed_share", [1.0, 0.6])
def test_step_with_a_separate_streaming_pass(hip, surveyed_share):
    """Big prefixes on pools with fresh known counts (B >= 64, launch bound N >= 603): the step kernel stops at the
    factor pairs and k_rank2v streams every covariance (two launches per step) -- against the one-launch form and the
    four-launches-per-slot form, bit for bit: decisions, known counts, states, covariances.  Filters with ragged reading
    counts (zero-filled pair rows), filters without readings in a step, and a pool in which 40 % of the filters have not
    surveyed the map (the 70 % rule then keeps to one launch per step: the forms must still agree)."""
    n, B, T = 320, 64, 6
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=777, half_extent=6.0, min_spacing=0.3, max_visible_dis=1.3, vmax=8,
                          v_cmd=1.0, w_cmd=0.6)
    log = synth.make_unknown_log(cfg)
    cnt = log.count.copy()
    cnt[2, ::5] = 0                      # some filters sit a step out
    world = log.world
    rng = np.random.default_rng(3)
    init = (world[None] + rng.normal(0.0, 0.005, size=(B, n, 2))).reshape(B, 2 * n)
    lm0 = np.full((2, B, 1), -1, dtype=np.int32)
    known0 = np.where(np.arange(B) < surveyed_share * B, n, 0).astype(np.int32)
    snap = []
    for mode in (1, 2, 0, -64):   # -64: the two-launch form with 64-row workgroups (k_rank2v with K staged in LDS)
        bt = hip.BatchEKF(B, n)
        bt.set_step_fused(abs(mode) if mode >= 0 else 1)
        if mode < 0:
            bt.set_tuning(rows_per_block=-mode)
        bt.upload_known_log(np.zeros((2, B, 2)), lm0, np.zeros((2, B, 1, 2)), init)
        bt.run_known()
        bt.set_known_counts(known0)
        bt.upload_unknown_log(log.twist, cnt, log.meas_xy)
        st = bt.run_unknown(0, T, time_kernels=True)
        snap.append((bt.decisions().copy(), bt.known_counts().copy(), [bt.state(b) for b in (0, 7, B - 1)],
                     [bt.cov(b) for b in (0, 7, B - 1)], bt.checksum(), st))
        bt.close()
    for other in (1, 2, 3):
        assert np.array_equal(snap[0][0], snap[other][0]) and np.array_equal(snap[0][1], snap[other][1])
        for k in range(3):
            assert np.array_equal(snap[0][2][k], snap[other][2][k]) and np.array_equal(snap[0][3][k], snap[other][3][k])
        assert np.allclose(np.array(snap[0][4]), np.array(snap[other][4]), rtol=1e-12)   # (atomic sums: order varies)
        assert snap[0][5]["corrections"] == snap[other][5]["corrections"] > 200
    assert snap[0][5]["rank2_launches"] == snap[1][5]["rank2_launches"] == T < snap[2][5]["rank2_launches"]


@pytest.mark.parametrize("B,n,mode", [(64, 320, 1), (6, 150, 1), (6, 150, 0), (5, 40, 1)])
def test_a_non_finite_reading_stays_inside_its_filter(hip, B, n, mode):
    """A NaN reading in ONE filter of an unknown-association run (the reference has no guards: it propagates silently
    through that filter's state and covariance) must leave every other filter of the pool bit-identical -- on the
    LDS-resident step kernel (n = 40), the one-launch step, the two-launch step (B = 64: fresh known counts) and the
    four-launch form."""
    T = 5
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=31 + n, half_extent=5.0, min_spacing=0.3, max_visible_dis=1.4, vmax=6,
                          v_cmd=1.0, w_cmd=0.6)
    log = synth.make_unknown_log(cfg)
    assert log.count[2, 1] >= 1
    rng = np.random.default_rng(8)
    init = (log.world[None] + rng.normal(0.0, 0.005, size=(B, n, 2))).reshape(B, 2 * n)
    lm0 = np.full((2, B, 1), -1, dtype=np.int32)
    res = []
    for poisoned in (True, False):
        meas = log.meas_xy.copy()
        if poisoned:
            meas[2, 1, 0, 1] = np.nan        # filter 1, step 2, first reading
        bt = hip.BatchEKF(B, n)
        bt.set_step_fused(mode)
        if n >= 150:     # big prefixes from the start: the map is surveyed first
            bt.upload_known_log(np.zeros((2, B, 2)), lm0, np.zeros((2, B, 1, 2)), init)
            bt.run_known()
            bt.set_known_counts(n)
        bt.upload_unknown_log(log.twist, log.count, meas)
        bt.run_unknown(0, T)
        res.append((bt.decisions().copy(), [bt.state(b) for b in range(B)], [bt.cov(b) for b in range(min(B, 8))]))
        bt.close()
    assert np.isnan(res[0][1][1]).any() or not np.array_equal(res[0][0][:, 1], res[1][0][:, 1])   # the reading mattered
    for b in range(B):
        if b == 1:
            continue
        assert np.array_equal(res[0][0][:, b], res[1][0][:, b]) and np.array_equal(res[0][1][b], res[1][1][b]), f"filter {b}"
        if b < 8:
            assert np.array_equal(res[0][2][b], res[1][2][b]), f"filter {b} covariance"


@pytest.mark.parametrize("k,surveyed_share", [(32, 1.0), (64, 0.6), (8, 1.0), (24, 1.0)])
def test_delayed_data_association_for_pools(hip, oracle, k, surveyed_share):
    """Delayed mode for pools' data_association() (ekf_slam.cpp:278-402): the pairs of a step stay pending ACROSS steps
    (jmax = 8 pairs per step and filter; k = 32 corrections per flush -> Sigma rewritten every 4 steps, k = 64 every 8,
    k = 8 every step, k = 24 every 3) while every reading is scored and corrected against the stored covariance minus ALL
    pending pairs.  Decisions and known counts identical to the eager run, states and covariances within 1e-9 of it and
    of the dense checker; ragged reading counts, filters that sit a step out, filters without a map yet (their first
    steps run on the LDS-resident path, which flushes)."""
    n, B, T = 150, 12, 11
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=4242, half_extent=5.0, min_spacing=0.3, max_visible_dis=1.4, vmax=8,
                          v_cmd=1.0, w_cmd=0.6)
    log = synth.make_unknown_log(cfg)
    cnt = log.count.copy()
    cnt[3, ::4] = 0                      # some filters sit a step out
    rng = np.random.default_rng(5)
    init = (log.world[None] + rng.normal(0.0, 0.005, size=(B, n, 2))).reshape(B, 2 * n)
    lm0 = np.full((2, B, 1), -1, dtype=np.int32)
    known0 = np.where(np.arange(B) < surveyed_share * B, n, 0).astype(np.int32)
    snap = []
    for mode in (0, k):
        bt = hip.BatchEKF(B, n)
        bt.upload_known_log(np.zeros((2, B, 2)), lm0, np.zeros((2, B, 1, 2)), init)
        bt.run_known()
        bt.set_known_counts(known0)
        bt.set_update_mode(mode)
        bt.upload_unknown_log(log.twist, cnt, log.meas_xy)
        st1 = bt.run_unknown(0, 5, time_kernels=True)
        st2 = bt.run_unknown(5, T, time_kernels=True)   # (a run boundary flushes)
        snap.append((bt.decisions().copy(), bt.known_counts().copy(), [bt.state(b) for b in range(B)],
                     [bt.cov(b) for b in (0, 5, B - 1)], st1, st2))
        bt.close()
    assert np.array_equal(snap[0][0], snap[1][0]) and np.array_equal(snap[0][1], snap[1][1])
    for b in range(B):
        assert np.abs(snap[0][2][b] - snap[1][2][b]).max() < 1e-9
    for i, b in enumerate((0, 5, B - 1)):
        assert_parity(snap[1][2][b], snap[1][3][i], snap[0][2][b], snap[0][3][i], FP64_TOL, f"delayed vs eager, filter {b}")
    if surveyed_share == 1.0:
        # passes over Sigma: one per step eagerly; one per floor(k / 8) steps (+ the run's closing flush) when delayed
        assert snap[0][4]["rank2_launches"] == 5 and snap[0][5]["rank2_launches"] == 6
        per = max(1, k // 8)
        assert snap[1][4]["rank2_launches"] == -(-5 // per) and snap[1][5]["rank2_launches"] == -(-6 // per)
    assert snap[0][4]["corrections"] + snap[0][5]["corrections"] == snap[1][4]["corrections"] + snap[1][5]["corrections"] > 300
    # against the dense checker (one surveyed filter: the checker replays the survey through its known-association path)
    b = 0
    o, known = oracle.OracleEKF(n, oracle.DENSE), np.zeros(n, dtype=np.uint8)
    o.prediction(0.0, 0.0); o.measurement_compact(init[b], lm0[0, b], np.zeros((1, 2)))
    o.prediction(0.0, 0.0); o.measurement_compact(init[b], lm0[1, b], np.zeros((1, 2)))
    known[:] = 1
    class _L: pass
    lg = _L(); lg.count, lg.twist, lg.meas_xy = cnt, log.twist, log.meas_xy
    o, known, d = _oracle_replay(oracle, lg, b, n, 0, T, o, known)
    assert np.array_equal(snap[1][0][:, b], d), "decisions differ from the checker"
    assert_parity(snap[1][2][b], snap[1][3][0], o.state, o.cov, FP64_TOL, "delayed pool association vs dense checker")


def test_delayed_data_association_with_the_symmetric_option(hip):
    """Pools' delayed data_association() with the symmetric option: the step kernel keeps reading columns and rows of the
    stored covariance, which the mirrored flush (k_flush_sym) leaves exactly symmetric outside its diagonal squares --
    decisions and known counts identical to the eager run, states and covariances within 1e-9."""
    n, B, T = 170, 9, 10
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=777, half_extent=5.0, min_spacing=0.3, max_visible_dis=1.4, vmax=8,
                          v_cmd=1.0, w_cmd=0.6)
    log = synth.make_unknown_log(cfg)
    rng = np.random.default_rng(6)
    init = (log.world[None] + rng.normal(0.0, 0.005, size=(B, n, 2))).reshape(B, 2 * n)
    lm0 = np.full((2, B, 1), -1, dtype=np.int32)
    snap = []
    for mode in (0, 24):
        bt = hip.BatchEKF(B, n)
        bt.upload_known_log(np.zeros((2, B, 2)), lm0, np.zeros((2, B, 1, 2)), init)
        bt.run_known()
        bt.set_known_counts(np.full(B, n, dtype=np.int32))
        bt.set_update_mode(mode, symmetric_gather=bool(mode))
        bt.upload_unknown_log(log.twist, log.count, log.meas_xy)
        bt.run_unknown(0, T)
        if mode:
            assert bt.form_counts()["flush_mirrored"] >= 3
        snap.append((bt.decisions().copy(), bt.known_counts().copy(), [bt.state(b) for b in range(B)], bt.cov(B - 1)))
        bt.close()
    assert np.array_equal(snap[0][0], snap[1][0]) and np.array_equal(snap[0][1], snap[1][1])
    for b in range(B):
        assert np.abs(snap[0][2][b] - snap[1][2][b]).max() < 1e-9
    assert_parity(snap[1][2][B - 1], snap[1][3], snap[0][2][B - 1], snap[0][3], FP64_TOL, "symmetric delayed vs eager")



@pytest.mark.parametrize("k,silent", [(32, (2,)), (64, (1, 2)), (64, (3, 6)), (32, (0,))])
def test_delayed_data_association_with_a_step_nobody_reads(hip, k, silent):
    """A delayed step in which NO filter of the pool has a reading (count[t, :] = 0) between two flushes: the step still
    predicts (ekf_slam.cpp:55-106), so the pending pairs and the cached 5 x 5 blocks must take At . At^T + Q before the next
    reading is scored (ekf_slam.cpp:300-309).  Decisions, known counts identical to the eager run, states and covariances
    within 1e-9 of it (round 3's advisor finding: the pool-wide silent step left the block cache one prediction behind)."""
    n, B, T = 150, 10, 10
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=991, half_extent=5.0, min_spacing=0.3, max_visible_dis=1.4, vmax=8,
                          v_cmd=1.0, w_cmd=0.6)
    log = synth.make_unknown_log(cfg)
    cnt = log.count.copy()
    for t in silent:
        cnt[t, :] = 0
    rng = np.random.default_rng(7)
    init = (log.world[None] + rng.normal(0.0, 0.005, size=(B, n, 2))).reshape(B, 2 * n)
    lm0 = np.full((2, B, 1), -1, dtype=np.int32)
    snap = []
    for mode in (0, k):
        bt = hip.BatchEKF(B, n)
        bt.upload_known_log(np.zeros((2, B, 2)), lm0, np.zeros((2, B, 1, 2)), init)
        bt.run_known()
        bt.set_known_counts(np.full(B, n, dtype=np.int32))
        bt.set_update_mode(mode)
        bt.upload_unknown_log(log.twist, cnt, log.meas_xy)
        bt.run_unknown(0, T)
        snap.append((bt.decisions().copy(), bt.known_counts().copy(), [bt.state(b) for b in range(B)],
                     [bt.cov(b) for b in (0, B - 1)]))
        bt.close()
    assert (snap[0][0] >= 0).sum() > 200                      # the run corrects
    assert np.array_equal(snap[0][0], snap[1][0]) and np.array_equal(snap[0][1], snap[1][1])
    for b in range(B):
        assert np.abs(snap[0][2][b] - snap[1][2][b]).max() < 1e-9, f"filter {b}"
    for i, b in enumerate((0, B - 1)):
        assert_parity(snap[1][2][b], snap[1][3][i], snap[0][2][b], snap[0][3][i], FP64_TOL, f"delayed vs eager, filter {b}")


@pytest.mark.parametrize("k,surveyed_share,symmetric", [(32, 1.0, False), (64, 0.6, False), (24, 1.0, False), (32, 1.0, True)])
def test_speculative_old_part_is_bit_identical(hip, k, surveyed_share, symmetric):
    """EKF_FORM_STEP_SPECULATE (pools' delayed data_association()): a launch in front of every step guesses each reading's
    winner and rebuilds "stored covariance minus the pairs of earlier steps" for the guessed landmarks once per step
    (k_pool_step_spec); the step kernel continues from it where its decision agrees (ekf_slam.cpp:300-309 decides, :331-390
    corrects) and rebuilds from scratch where it does not -- new landmarks, readings between the gates, filters still
    discovering their map.  Same operations in the same order: decisions, known counts, states and covariances must be
    BIT-identical to the run with the form off; ragged reading counts, filters that sit steps out, a run boundary."""
    n, B, T = 150, 10, 12
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=6161, half_extent=5.0, min_spacing=0.3, max_visible_dis=1.4, vmax=8,
                          v_cmd=1.0, w_cmd=0.6)
    log = synth.make_unknown_log(cfg)
    cnt = log.count.copy()
    cnt[4, ::3] = 0
    meas = log.meas_xy.copy()
    meas[6, :, 0] += (0.2, -0.15)         # off-landmark readings: the guess (nearest landmark) and the decision may part
    rng = np.random.default_rng(9)
    init = (log.world[None] + rng.normal(0.0, 0.005, size=(B, n, 2))).reshape(B, 2 * n)
    lm0 = np.full((2, B, 1), -1, dtype=np.int32)
    known0 = np.where(np.arange(B) < surveyed_share * B, n, 0).astype(np.int32)
    snap = []
    for forms in (hip.FORMS_DEFAULT, hip.FORMS_DEFAULT & ~hip.FORM_STEP_SPECULATE):
        bt = hip.BatchEKF(B, n)
        bt.set_forms(forms)
        bt.upload_known_log(np.zeros((2, B, 2)), lm0, np.zeros((2, B, 1, 2)), init)
        bt.run_known()
        bt.set_known_counts(known0)
        bt.set_update_mode(k, symmetric_gather=symmetric)
        bt.upload_unknown_log(log.twist, cnt, meas)
        bt.run_unknown(0, 7); bt.run_unknown(7, T)
        snap.append((bt.decisions().copy(), bt.known_counts().copy(), [bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)]))
        bt.close()
    assert np.array_equal(snap[0][0], snap[1][0]) and np.array_equal(snap[0][1], snap[1][1])
    assert (snap[0][0] >= 0).sum() > 300
    for b in range(B):
        assert np.array_equal(snap[0][2][b], snap[1][2][b]) and np.array_equal(snap[0][3][b], snap[1][3][b]), f"filter {b}"
