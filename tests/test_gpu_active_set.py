"""-m gpu: active-set covariance update (opt-in): the eager correction streams only the rows of the
touched set.  Must be BIT-identical to the dense path (every skipped row has K = 0 exactly)."""
import numpy as np
import pytest

from ekf_slam_ml_amd import synth
from parity import FP64_TOL, assert_parity

pytestmark = pytest.mark.gpu


def test_active_set_against_the_cpu_checker(hip, oracle):
    """Direct comparison with the CPU restatement (not only with the eager HIP path): single filter n = 200 against
    the dense-literal checker, and a pool at n = 400 against the structured one."""
    T = 25
    log = synth.make_known_log(synth.config2(steps=T))
    f = hip.EKF_SLAM(200)
    f.set_active_set(True)
    o = oracle.OracleEKF(200, oracle.DENSE)
    for t in range(T):
        sensor, vis = log.expand_step(t)
        f.prediction(log.twist[t, 0]); o.prediction(*log.twist[t, 0])
        f.measurement(sensor, vis);    o.measurement(sensor, vis)
    assert_parity(f.state, f.cov, o.state, o.cov, FP64_TOL, "active-set single filter vs dense checker")
    f.close()

    cfg = synth.config5(filters=6, steps=10, n=400)
    blog = synth.make_known_log(cfg)
    bt = hip.BatchEKF(6, 400)
    bt.set_active_set(True)
    bt.upload_known_log(blog.twist, blog.lm_idx, blog.z_xy, blog.init_xy)
    bt.run_known()
    st, cv, _ = oracle.batch_run_known(blog, oracle.STRUCTURED, want_cov=True, fast=False)
    for b in range(6):
        assert_parity(bt.state(b), bt.cov(b), st[b], cv[b], FP64_TOL, f"active-set pool filter {b} vs structured checker")
    assert bt.touched().max() < 400   # the sparse path really was taken
    bt.close()


def test_single_filter_active_set_bitwise(hip):
    log = synth.make_known_log(synth.config2(steps=40))
    outs = []
    for on in (False, True):
        f = hip.EKF_SLAM(200)
        f.set_active_set(on)
        for t in range(40):
            sensor, vis = log.expand_step(t)
            f.prediction(log.twist[t, 0]); f.measurement(sensor, vis)
            if t == 20:   # interleave an association call and a clone: the touched set must follow
                k = np.ones(200, dtype=np.uint8)
                f.data_association(log.z_xy[t, 0, :2], k)
                g = f.clone(); f.close(); f = g
        outs.append((f.state, f.cov))
        f.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_set_cov_marks_everything_touched(hip):
    rng = np.random.default_rng(2)
    n = 30
    a = rng.normal(size=(3 + 2 * n, 3 + 2 * n))
    cov = a @ a.T + np.eye(3 + 2 * n)                 # fully dense covariance supplied by the caller
    st = rng.normal(size=3 + 2 * n)
    sensor = rng.normal(size=2 * n) + 2.0
    outs = []
    for on in (False, True):
        f = hip.EKF_SLAM(n)
        f.set_small_map_path(False)
        f.set_active_set(on)
        f.state, f.cov = st, cov
        f.landmark_init_flag = True
        vis = np.zeros(n, dtype=np.uint8); vis[[3, 7, 19]] = 1
        f.prediction((0.1, 0.05))
        f.measurement(sensor, vis)
        outs.append((f.state, f.cov))
        f.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_batch_active_set_bitwise_n1000(hip):
    cfg = synth.config5(filters=10, steps=12, n=1000)
    log = synth.make_known_log(cfg)
    res = []
    for on in (False, True):
        bt = hip.BatchEKF(10, 1000)
        bt.set_active_set(on)
        bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
        bt.run_known(0, 5)
        st = bt.run_known(5, 12, time_kernels=True)
        res.append(([bt.state(b) for b in range(10)], [bt.cov(b) for b in (0, 4, 9)], bt.checksum(), st))
        bt.close()
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)
    assert res[1][3]["rank2_ms"] < res[0][3]["rank2_ms"]   # and it must actually be cheaper


def test_batch_active_set_with_device_generated_log(hip):
    cfg = synth.config5(filters=9, steps=10, n=400)
    world = synth.make_world(cfg.n, cfg.half_extent, cfg.min_spacing, cfg.world_seed)
    res = []
    for on in (False, True):
        bt = hip.BatchEKF(9, 400)
        bt.set_active_set(on)
        bt.simulate_known_log(cfg, world)
        bt.run_known()
        res.append([(bt.state(b), bt.cov(b)) for b in range(9)])
        bt.close()
    for (s0, c0), (s1, c1) in zip(*res):
        assert np.array_equal(s0, s1) and np.array_equal(c0, c1)


def test_second_log_on_used_filters(hip):
    """A new log uploaded onto filters that already carry touched landmarks: the grid bound of the
    active-set kernel must cover the union of old and new touched sets."""
    cfg1 = synth.config5(filters=4, steps=6, n=300)
    cfg2 = synth.config5(filters=4, steps=6, n=300)
    cfg2.seed = 777                      # different noise, and ...
    cfg2.v_cmd, cfg2.w_cmd = -0.8, 0.3   # ... a different path: other landmarks come into view
    l1, l2 = synth.make_known_log(cfg1), synth.make_known_log(cfg2)
    res = []
    for on in (False, True):
        bt = hip.BatchEKF(4, 300)
        bt.set_active_set(on)
        bt.upload_known_log(l1.twist, l1.lm_idx, l1.z_xy, l1.init_xy)
        bt.run_known()
        bt.upload_known_log(l2.twist, l2.lm_idx, l2.z_xy, l2.init_xy)   # no reset in between
        bt.run_known()
        res.append([(bt.state(b), bt.cov(b)) for b in range(4)] + [bt.touched()])
        bt.close()
    for b in range(4):
        assert np.array_equal(res[0][b][0], res[1][b][0]) and np.array_equal(res[0][b][1], res[1][b][1])
    assert np.array_equal(res[0][4], res[1][4]) and res[1][4].max() >= 3
