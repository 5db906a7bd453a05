"""-m gpu: on-device Monte-Carlo log generator + consistency statistics.
The device generator must reproduce the host generator (ekf_slam_ml_amd/synth.py: same noise model, same
random-number addressing): identical landmark slots, values to rounding.  The filter run on the
device-made log is checked against the CPU checker fed with the SAME (downloaded) log."""
import numpy as np
import pytest

from ekf_slam_ml_amd import synth
from parity import FP64_TOL, assert_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [
    synth.SimConfig(n=40, steps=25, filters=7, seed=123, half_extent=2.0, min_spacing=0.2, max_visible_dis=0.8, vmax=5),
    synth.config5(filters=9, steps=8, n=300, first_filter_id=1000),
    synth.SimConfig(n=12, steps=6, filters=3, seed=5, v_cmd=0.2, w_cmd=0.0, max_visible_dis=2.0, vmax=12),
])
def test_device_log_equals_host_log(hip, cfg):
    host = synth.make_known_log(cfg)
    bt = hip.BatchEKF(cfg.filters, cfg.n)
    bt.simulate_known_log(cfg, host.world)
    tw, li, zz, ii, tp = bt.download_log()
    assert np.array_equal(li, host.lm_idx)                 # same landmarks in the same slots
    assert np.abs(tw - host.twist).max() < 1e-12
    assert np.abs(tp - host.true_pose).max() < 1e-11
    assert np.abs(zz - host.z_xy).max() < 1e-11 and np.abs(ii - host.init_xy).max() < 1e-11
    st = bt.run_known()
    assert st["corrections"] == host.corrections
    bt.close()


def test_filter_on_device_log_vs_checker_and_mc_stats(hip, oracle):
    cfg = synth.config5(filters=64, steps=30, n=100)
    cfg.max_visible_dis, cfg.vmax = 3.0, 4
    world = synth.make_world(cfg.n, cfg.half_extent, cfg.min_spacing, cfg.world_seed)
    bt = hip.BatchEKF(cfg.filters, cfg.n)
    bt.simulate_known_log(cfg, world)
    tw, li, zz, ii, tp = bt.download_log()
    bt.run_known()
    log = synth.KnownLog(cfg, world, tw, li, zz, ii, tp)
    st, cv, _ = oracle.batch_run_known(log, oracle.STRUCTURED, want_cov=True, fast=False)
    for b in (0, 17, 63):
        assert_parity(bt.state(b), bt.cov(b), st[b], cv[b], FP64_TOL, f"filter {b}")
    s = bt.mc_stats(cfg.steps - 1)
    # host recomputation of the same statistics from the checker's states / covariances
    e = st[:, :3] - tp[-1]
    e[:, 0] = np.angle(np.exp(1j * e[:, 0]))
    nees = np.array([e[b] @ np.linalg.solve(cv[b][:3, :3], e[b]) for b in range(cfg.filters)])
    assert abs(s["nees_mean"] - nees.mean()) < 1e-6 * max(1.0, nees.mean())
    assert abs(s["rmse_xy"] - np.sqrt((e[:, 1:] ** 2).sum(axis=1).mean())) < 1e-9
    assert abs(s["rmse_theta"] - np.sqrt((e[:, 0] ** 2).mean())) < 1e-9
    assert abs(s["frac_nees_below_95pct"] - (nees < 7.815).mean()) < 1e-12
    # the filter tracks the truth; R = 0.01 is far above the simulated sensor noise (0.005^2), so the filter
    # is conservative: NEES well below its 3-dof expectation is the expected outcome, not a bug
    assert s["rmse_xy"] < 0.3 and s["rmse_theta"] < 0.3 and s["nees_mean"] < 10.0
    bt.close()


def test_mc_stats_need_a_simulated_log(hip):
    log = synth.make_known_log(synth.config5(filters=2, steps=3, n=10))
    bt = hip.BatchEKF(2, 10)
    bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
    bt.run_known()
    with pytest.raises(hip.EkfError):
        bt.mc_stats(2)
    bt.close()
