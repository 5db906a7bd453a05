"""-m gpu: measurement() as two launches per call (ekf_callfused.hip): the gains K_v and rows G_v = H_v Sigma of all
the call's corrections from two thin panels of Sigma, then ONE read-modify-write pass in which every element takes its
V rank-2 corrections in order.  Must be BIT-identical to the launch-per-landmark path (single filters and pools, ragged
visible counts, calls longer than one pass, calls without any visible landmark) and agree with the CPU checker."""
import numpy as np
import pytest

from ekf_slam_ml_amd import synth
from parity import FP64_TOL, assert_parity

pytestmark = pytest.mark.gpu


def _replay(f, log, t0, t1):
    for t in range(t0, t1):
        sensor, vis = log.expand_step(t)
        f.prediction(log.twist[t, 0])
        f.measurement(sensor, vis)


def _single(hip, n, call_fused):
    f = hip.EKF_SLAM(n)
    f.set_call_fused(call_fused)
    return f


def test_call_fused_config1_bitwise_and_vs_checker(hip, oracle):
    T = 40
    log = synth.make_known_log(synth.config2(steps=T))
    outs = []
    for cf in (True, False):
        f = _single(hip, 200, cf)
        _replay(f, log, 0, 15)
        g = f.clone()
        f.close()
        _replay(g, log, 15, T)
        outs.append((g.state, g.cov))
        g.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    o = oracle.OracleEKF(200, oracle.DENSE)
    for t in range(10):
        sensor, vis = log.expand_step(t)
        o.prediction(*log.twist[t, 0]); o.measurement(sensor, vis)
    f = _single(hip, 200, True)
    _replay(f, log, 0, 10)
    assert_parity(f.state, f.cov, o.state, o.cov, FP64_TOL, "call-fused measurement vs dense checker")
    f.close()


@pytest.mark.parametrize("n,vmax", [(60, 3), (60, 60), (101, 9), (333, 17), (700, 8), (1000, 11)])
def test_call_fused_shapes_and_long_calls(hip, n, vmax):
    """V from 0 to beyond one pass (8 corrections): several passes per call, the pose of the call is captured once."""
    cfg = synth.SimConfig(n=n, steps=9, filters=1, seed=300 + n + vmax, half_extent=3.0, min_spacing=0.12,
                          max_visible_dis=3.0 if vmax > 8 else 1.2, vmax=vmax)
    log = synth.make_known_log(cfg)
    counts = (log.lm_idx[:, 0] >= 0).sum(axis=1)
    assert counts.max() >= min(vmax, 8)
    res = []
    for cf in (True, False):
        f = _single(hip, n, cf)
        _replay(f, log, 0, cfg.steps)
        vis0 = np.zeros(n, dtype=np.uint8)
        f.prediction((0.02, 0.01)); f.measurement(log.expand_step(3)[0], vis0)     # a call without visible landmarks
        res.append((f.state, f.cov))
        f.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.all(np.isfinite(res[0][1]))


def test_call_fused_interleaved_with_association_and_snapshots(hip):
    n = 150
    cfg = synth.SimConfig(n=n, steps=20, filters=1, seed=78, half_extent=2.5, min_spacing=0.2, max_visible_dis=1.0, vmax=8)
    log = synth.make_known_log(cfg)
    res = []
    for cf in (True, False):
        f = _single(hip, n, cf)
        for t in range(cfg.steps):
            sensor, vis = log.expand_step(t)
            f.prediction(log.twist[t, 0])
            if t % 4 == 3:
                k = np.ones(n, dtype=np.uint8)
                f.data_association(log.z_xy[t, 0, :3], k)
            else:
                f.measurement(sensor, vis)
            if t == 9:
                st, cv = f.state, f.cov
                f.state, f.cov = st, cv
            if t == 13:
                f.set_call_fused(not cf)
            if t == 16:
                f.set_call_fused(cf)
        res.append((f.state, f.cov))
        f.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("B,n,vmax", [(7, 400, 5), (5, 130, 12), (10, 1000, 2), (6, 401, 7), (3, 640, 16)])
def test_call_fused_pool_bitwise_and_vs_checker(hip, oracle, B, n, vmax):
    """Pools: ragged visible counts across filters (some filters sit a pass out), several passes per call, split runs."""
    cfg = synth.SimConfig(n=n, steps=8, filters=B, seed=900 + n, half_extent=4.0, min_spacing=0.15,
                          max_visible_dis=1.5 if vmax <= 5 else 3.0, vmax=vmax)
    if n == 1000:
        cfg = synth.config5(filters=B, steps=8, n=n)
    log = synth.make_known_log(cfg)
    counts = (log.lm_idx >= 0).sum(axis=2)
    res = []
    for cf in (True, False):
        bt = hip.BatchEKF(B, n)
        bt.set_call_fused(cf)
        if n in (401, 640):
            bt.set_tuning(rows_per_block=64 if n == 401 else 32)   # the big-pool form of the pass: K values staged in LDS
        bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
        bt.run_known(0, 3)
        st = bt.run_known(3, cfg.steps, time_kernels=True)
        res.append(([bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)], st))
        bt.close()
    for b in range(B):
        assert np.array_equal(res[0][0][b], res[1][0][b]), f"filter {b} state"
        assert np.array_equal(res[0][1][b], res[1][1][b]), f"filter {b} covariance"
    assert res[0][2]["corrections"] == res[1][2]["corrections"] == int(counts[3:].sum())
    assert res[0][2]["rank2_launches"] <= res[1][2]["rank2_launches"]
    if n <= 400:
        sto, cvo, _ = oracle.batch_run_known(log, oracle.STRUCTURED, want_cov=True, fast=False)
        for b in range(B):
            assert_parity(res[0][0][b], res[0][1][b], sto[b], cvo[b], FP64_TOL, f"call-fused pool filter {b} vs checker")


def test_call_fused_pool_on_device_log_full_size(hip):
    """BASELINE.json configs[4]'s shape at reduced B, inputs simulated on the device: one pass per step instead of two."""
    cfg = synth.config5(filters=24, steps=7, n=1000)
    world = synth.make_world(cfg.n, cfg.half_extent, cfg.min_spacing, cfg.world_seed)
    res = []
    for cf in (True, False):
        bt = hip.BatchEKF(24, 1000)
        bt.set_call_fused(cf)
        bt.simulate_known_log(cfg, world)
        st = bt.run_known(0, 7, time_kernels=True)
        res.append((bt.checksum(), [bt.state(b) for b in (0, 11, 23)], bt.cov(23), st))
        bt.close()
    assert np.allclose(np.array(res[0][0]), np.array(res[1][0]), rtol=1e-12)   # (the digest sums with atomics: order varies)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)
    assert np.array_equal(res[0][2], res[1][2])
    assert res[0][3]["rank2_launches"] * 2 == res[1][3]["rank2_launches"]   # V = 2 corrections per call


@pytest.mark.parametrize("n,vmax", [(750, 6), (800, 12), (1000, 8), (470, 7)])
def test_call_fused_data_association_bitwise_and_vs_checker(hip, oracle, n, vmax):
    """data_association() beyond the LDS-resident path with Sigma streamed once per call (ekf_assocfused.hip: one launch
    per reading against the stored covariance minus the call's pending pairs, one k_rank2v pass per 8 readings) against
    the per-reading chain (k_maha, k_assoc_decide, k_gain, k_rank2: four launches per reading) bit for bit, and against the checker:
    new landmarks, matched landmarks, dropped readings (1 <= d < 10), calls longer than one pass, measurement() in between."""
    T = 14 if n < 1000 else 12
    cfg = synth.SimConfig(n=n, steps=T, filters=1, seed=77 + n, half_extent=12.0, min_spacing=0.3,
                          max_visible_dis=1.2 if vmax <= 8 else 2.0, vmax=vmax, v_cmd=1.0, w_cmd=0.5)
    log = synth.make_unknown_log(cfg)
    rng = np.random.default_rng(5)
    res = []
    for cf in (True, False):
        f = _single(hip, n, cf)
        f.set_active_prefix(False)    # full-width corrections: beyond the one-workgroup form from the first reading on
        known = np.zeros(n, dtype=np.uint8)
        decs = []
        r = np.random.default_rng(9)
        for t in range(T):
            J = int(log.count[t, 0])
            m = log.meas_xy[t, 0, :J].copy()
            if t % 5 == 2 and J > 1:
                m[0] += (0.12, -0.09)     # an off-landmark reading: lands between the gates now and then (dropped)
            f.prediction(log.twist[t, 0])
            decs.append(f.data_association(m, known).copy())
            if t == T // 2:               # a known-association call on what has been discovered so far
                vis = known.copy(); vis[int(known.sum()) // 2:] = 0
                sensor = r.normal(0, 1.0, size=2 * n)
                f.landmark_init_flag = True
                f.prediction((0.01, 0.01)); f.measurement(sensor * 0 + np.tile(m[0], n), vis)
        res.append((f.state, f.cov, decs, known.copy()))
        f.close()
    assert all(np.array_equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    assert np.array_equal(res[0][3], res[1][3]) and res[0][3].sum() >= 6
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    if n <= 800 and vmax <= 7:   # and the checker (decisions first)
        o = oracle.OracleEKF(n, oracle.STRUCTURED)
        ko = np.zeros(n, dtype=np.uint8)
        r = np.random.default_rng(9)
        for t in range(T):
            J = int(log.count[t, 0])
            m = log.meas_xy[t, 0, :J].copy()
            if t % 5 == 2 and J > 1:
                m[0] += (0.12, -0.09)
            o.prediction(*log.twist[t, 0])
            assert np.array_equal(o.data_association(m, ko), res[0][2][t]), f"step {t}"
            if t == T // 2:
                vis = ko.copy(); vis[int(ko.sum()) // 2:] = 0
                r.normal(0, 1.0, size=2 * n)
                o.set_init_flag(1)
                o.prediction(0.01, 0.01); o.measurement(np.tile(m[0], n), vis)
        assert_parity(res[0][0], res[0][1], o.state, o.cov, FP64_TOL, "call-fused data_association vs checker")


@pytest.mark.parametrize("call_fused", [True, False])
def test_a_filter_full_of_nan_leaves_its_pool_neighbours_alone(hip, call_fused):
    """Filters of a pool share nothing (ekf_slam.hpp:61-65): one filter fed a non-finite reading turns to NaN (the
    reference has no guards: arithmetic faults propagate silently); its neighbours must come out bit-identical to a run
    in which that filter is healthy -- factor rows, counts and passes are per filter."""
    B, n, T = 5, 130, 6
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=17, half_extent=4.0, min_spacing=0.2, max_visible_dis=2.0, vmax=4)
    log = synth.make_known_log(cfg)
    assert (log.lm_idx[1, 2] >= 0).any()
    res = []
    for poisoned in (True, False):
        z = log.z_xy.copy()
        if poisoned:
            z[1, 2, 0, 0] = np.nan      # filter 2, step 1, first visible landmark
        bt = hip.BatchEKF(B, n)
        bt.set_call_fused(call_fused)
        bt.upload_known_log(log.twist, log.lm_idx, z, log.init_xy)
        bt.run_known(0, T)
        res.append(([bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)]))
        bt.close()
    assert np.isnan(res[0][0][2]).any() and np.isnan(res[0][1][2]).any() and not np.isnan(res[1][0][2]).any()
    for b in (0, 1, 3, 4):
        assert np.array_equal(res[0][0][b], res[1][0][b]) and np.array_equal(res[0][1][b], res[1][1][b]), f"filter {b}"


@pytest.mark.parametrize("n,J", [(300, 100), (1000, 70), (150, 130)])
def test_calls_with_more_than_64_readings(hip, n, J):
    """data_association() calls that outgrow the 64-reading capacity (the decision block and its mapped host copy are
    re-allocated) and span many passes of 8 readings: the call-fused forms (whole call / one launch per reading, decisions
    published to the host's copy by the deciding launches themselves) against the per-reading form, which ends with the
    separate publishing launch -- decisions, known_list, state and covariance bit for bit."""
    world = synth.make_world(n, 10.0, 0.5, 5)
    outs = []
    for cf in (True, False):
        f = hip.EKF_SLAM(n)
        f.set_call_fused(cf)
        known = np.zeros(n, dtype=np.uint8)
        rng = np.random.default_rng(7)
        decs = []
        for _ in range(4):
            f.prediction(np.array([0.02, 0.1]))
            idx = rng.choice(n, size=J, replace=False)
            decs.append(np.array(f.data_association(world[idx] + rng.normal(0, 0.01, size=(J, 2)), known)).copy())
        outs.append((np.concatenate(decs), known.copy(), f.state.copy(), f.cov.copy()))
        f.close()
    assert outs[0][1].sum() > 100
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)



def test_two_handles_driven_from_two_host_threads(hip, oracle):
    """One handle is not thread-safe, but two handles are independent (own stream, own staging ring, own mapped result
    block, thread-local error text): two host threads drive one EKF_SLAM object each through 50 ticks of the node loop
    at the same time -- one known association (prediction + measurement, slam.cpp:433-434), one unknown
    (prediction + data_association, unknown_data_assoc.cpp:414-415) -- and each ends where the checker ends.
    ctypes releases the GIL around every C-ABI call, so the calls of the two threads do overlap."""
    import threading
    T = 50
    klog = synth.make_known_log(synth.config2(steps=T))                       # n = 200, two launches per call
    ucfg = synth.SimConfig(n=150, steps=T, filters=1, seed=81, half_extent=5.0, min_spacing=0.3, max_visible_dis=1.4, vmax=8,
                           v_cmd=1.0, w_cmd=0.6)
    ulog = synth.make_unknown_log(ucfg)
    out, errs = {}, []
    gate = threading.Barrier(2)

    def known_side():
        try:
            f = hip.EKF_SLAM(200)
            gate.wait()
            _replay(f, klog, 0, T)
            out["known"] = (f.state, f.cov)
            f.close()
        except Exception as e:   # (reported by the main thread)
            errs.append(e)

    def unknown_side():
        try:
            f = hip.EKF_SLAM(150)
            known = np.zeros(150, dtype=np.uint8)
            dec = []
            gate.wait()
            for t in range(T):
                f.prediction(ulog.twist[t, 0])
                dec.append(f.data_association(ulog.meas_xy[t, 0, :ulog.count[t, 0]], known).copy())
            out["unknown"] = (f.state, f.cov, known, dec)
            f.close()
        except Exception as e:
            errs.append(e)

    th = [threading.Thread(target=known_side), threading.Thread(target=unknown_side)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    o = oracle.OracleEKF(200, oracle.STRUCTURED)
    for t in range(T):
        sensor, vis = klog.expand_step(t)
        o.prediction(*klog.twist[t, 0]); o.measurement(sensor, vis)
    assert_parity(out["known"][0], out["known"][1], o.state, o.cov, FP64_TOL, "known side, two threads")
    o = oracle.OracleEKF(150, oracle.STRUCTURED)
    known = np.zeros(150, dtype=np.uint8)
    for t in range(T):
        o.prediction(*ulog.twist[t, 0])
        d = o.data_association(ulog.meas_xy[t, 0, :ulog.count[t, 0]], known)
        assert np.array_equal(d, out["unknown"][3][t]), f"decisions of step {t}"
    assert np.array_equal(known, out["unknown"][2])
    assert_parity(out["unknown"][0], out["unknown"][1], o.state, o.cov, FP64_TOL, "unknown side, two threads")


def test_device_error_word_fails_the_handle_and_only_that_handle(hip):
    """A kernel that cannot go on (a bounded in-kernel hand-off that never arrives, ekf_callfused.hip await()) raises the
    pool's device error word in mapped host memory instead of continuing with stale operands.  The test hook raises it the
    same way: every later entry point of THAT handle fails with EKF_ERR_HIP (sticky), a second handle is untouched."""
    f, g = hip.EKF_SLAM(150), hip.EKF_SLAM(150)
    f.prediction((0.01, 0.05)); g.prediction((0.01, 0.05))
    assert hip.load().ekf_test_raise_device_error(f._h) == 0          # the raising call itself returns EKF_OK
    for call in (lambda: f.prediction((0.0, 0.01)), lambda: f.state, lambda: f.cov, lambda: f.sync(),
                 lambda: f.data_association(np.array([[1.0, 0.2]]), np.zeros(150, dtype=np.uint8))):
        with pytest.raises(hip.EkfError) as e:
            call()
        assert e.value.status == 3 and "device-side error" in str(e.value)   # EKF_ERR_HIP
    g.prediction((0.0, 0.01))
    assert np.isfinite(g.state).all() and g.getStateX() > 0.05
    f.close(); g.close()
