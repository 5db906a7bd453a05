"""CPU (not gpu): host logic -- the C-ABI library loads and exports every symbol include/ekfslam.h
declares, fails loudly without a device, and the synthetic-log generator is deterministic."""
import ctypes
import os
import re

import numpy as np
import pytest

from ekf_slam_ml_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def test_library_exports_every_declared_symbol():
    _built()
    hdr = open(os.path.join(ROOT, "include", "ekfslam.h")).read()
    declared = sorted(set(re.findall(r"^(?:ekf_status|const char\*|void|int)\s+(ekf_[a-z0-9_]+)\s*\(", hdr, re.M)))
    assert len(declared) >= 30
    assert sorted(capi.SYMBOLS) == declared, "capi.SYMBOLS and include/ekfslam.h drifted apart"
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libekfslam_hip.so does not export {name}"


def test_leading_dimension_rule():
    """ekf_leading_dimension (no device needed): N = 3 + 2 n rounded up to 16 doubles, or to 256 -- rows on 2-KB boundaries --
    where that adds at most 1/32 of a row; the values DESIGN.md section 2 and INTEGRATION.md quote."""
    _built()
    assert capi.leading_dimension(1000) == 2048 and capi.leading_dimension(5000) == 10240
    assert capi.leading_dimension(200) == 416 and capi.leading_dimension(20) == 48 and capi.leading_dimension(0) == 16
    for n in range(0, 6000, 7):
        N, ld = 3 + 2 * n, capi.leading_dimension(n)
        assert ld >= N and ld % 16 == 0
        wide = (N + 255) // 256 * 256
        assert ld == (wide if (wide - N) * 32 <= N else (N + 15) // 16 * 16)


def test_no_cpu_fallback():
    """Without a HIP device the product must refuse to run (no silent CPU path)."""
    _built()
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.EkfError) as e:
        capi.EKF_SLAM(5)
    assert e.value.status == 2  # EKF_ERR_NO_DEVICE
    with pytest.raises(capi.EkfError):
        capi.BatchEKF(2, 5)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ekf_slam_ml_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "libekf_oracle" not in src and "ekf_oracle.c" not in src.replace("oracle/ekf_oracle.c mode 1", ""), f


def test_default_params_are_the_reference_constants():
    _built()
    p = capi.default_params()
    # ekf_slam.cpp:32, :41-43, :174-175, :293, :330, :79
    assert (p.sigma0_landmark, p.q_pose, p.r_meas, p.gate_new, p.gate_update, p.straight_eps) == \
        (100.0, 0.0001, 0.01, 10.0, 1.0, 0.000001)


def test_splitmix_known_answers():
    # splitmix64 reference outputs for seed 0: first three values of the canonical generator
    x = np.uint64(0)
    outs = []
    for _ in range(3):
        outs.append(int(synth.splitmix64(x)))
        with np.errstate(over="ignore"):
            x = x + np.uint64(0x9E3779B97F4A7C15)
    assert outs == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_logs_are_deterministic_and_well_formed():
    a = synth.make_known_log(synth.config1(steps=30))
    b = synth.make_known_log(synth.config1(steps=30))
    for k in ("twist", "lm_idx", "z_xy", "init_xy", "world"):
        assert np.array_equal(getattr(a, k), getattr(b, k))
    assert (a.lm_idx[0] == -1).all()  # first call: visible_list all false (slam.cpp:315-327)
    for t in range(30):
        idx = a.lm_idx[t, 0]
        v = idx[idx >= 0]
        assert (np.diff(v) > 0).all() and (idx[len(v):] == -1).all()
    assert np.allclose(a.world[:10, 0], synth.TUBE_X) and np.allclose(a.world[:10, 1], synth.TUBE_Y)


def test_per_filter_streams_are_independent_and_shardable():
    """Filter g's inputs depend on its GLOBAL id only: a shard [4,8) of an 8-filter job equals filters
    4..7 of the unsharded job (this is what makes multi-GPU sharding a pure partition)."""
    full = synth.make_known_log(synth.config5(filters=8, steps=5, n=40))
    shard = synth.make_known_log(synth.config5(filters=4, steps=5, n=40, first_filter_id=4))
    assert np.array_equal(full.twist[:, 4:], shard.twist)
    assert np.array_equal(full.lm_idx[:, 4:], shard.lm_idx)
    assert np.array_equal(full.z_xy[:, 4:], shard.z_xy)
    assert np.array_equal(full.init_xy[4:], shard.init_xy)
    assert not np.array_equal(full.twist[:, 0], full.twist[:, 1])
    assert (full.lm_idx[1:] >= 0).all()  # exactly V = 2 readings per step in configs[4]


def test_unknown_log_shape():
    cfg = synth.config1(steps=20)
    log = synth.make_unknown_log(cfg)
    assert log.meas_xy.shape == (20, 1, cfg.vmax, 2) and (log.count <= cfg.vmax).all() and log.count.sum() > 20


def test_committed_bench_line_keeps_the_contract():
    """profiles/rNN/bench_n1.json (the latest round's) is the line bench.py printed on the GPU box: every field of the
    driver's contract must be there, with the roofline and cpu_baseline objects, and the numbers must be mutually
    consistent."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]", "bench_n1.json")))[-1]
    line = [l for l in open(path) if l.startswith("{")][0]
    d = json.loads(line)
    # The driver keeps an 8 KB tail of the run: the whole line must fit it (round 3's 14.4 KB line lost the delayed leg
    # -- the >= 1e6 figure -- and most side legs there).  bench.py asserts the same budget before it prints.
    assert len(line) < 7500, len(line)
    for k in ("delayed_update", "call_fused_update", "configs_1", "configs_2", "configs_3"):
        assert k in d, k
    assert {"value", "frac_of_8TBps", "max_abs_state_diff_vs_eager", "max_abs_state_diff_vs_cpu_port"} <= set(d["delayed_update"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 * r["frac"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # whole-job throughput = corrections of the timed steps / wall time
    corr = d["config"]["filters_per_gpu"] * d["config"]["corrections_per_filter_step"] * d["steps"]
    assert abs(d["value"] - corr / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]


def test_scan_generator_addressing():
    """The host twin of the device lidar: noise is a pure function of (seed, filter id, step, beam); the default filter
    id is the scan index; geometry is independent of the noise stream."""
    from ekf_slam_ml_amd import synth
    poses = np.array([[0.0, 0.0, 0.0], [1.0, 0.2, -0.3], [-2.0, -0.4, 0.5]])
    a = synth.make_scans(poses, seed=5)
    assert np.array_equal(a, synth.make_scans(poses, seed=5, fid=np.arange(3), step=0))
    b = synth.make_scans(poses, seed=5, fid=[7, 8, 9], step=4)
    clean = synth.make_scans(poses, seed=5, range_std=0.0)
    assert not np.array_equal(a, b) and np.abs(a - clean).max() < 0.05 and np.abs(b - clean).max() < 0.05
    assert np.array_equal(synth.make_scans(poses[1:2], seed=5, fid=[8], step=4)[0], b[1])   # scans are independent
    assert a.shape == (3, 360) and (clean <= 3.5).all() and (clean > 0).all()
    # a tube straight ahead of the first pose is hit at (distance - radius)
    tube = np.array([[1.0, 0.0]])
    r = synth.make_scans(poses[:1], world=tube, range_std=0.0, border=10.0)[0]
    assert abs(r[0] - (1.0 - synth.TUBE_RADIUS)) < 1e-12 and r[180] == 3.5


def test_reference_scan_model_agrees_with_ray_geometry_away_from_tubes():
    """ekf_lidar_params.model 1 = publishScan's own procedure (nurtlesim/src/tube_world.cpp:496-570: a bearing window of
    2 atan2(radius, range_min) around every tube, then the nearer intersection of the LINE through the robot and the beam's end
    point with the tube's circle) against model 0 (clean ray geometry): the same ranges to rounding for every pose that has
    no tube closer than radius / sin(atan2(radius, range_min)) = 0.142 m -- closer tubes subtend more than the window and the
    reference clips them, which is the one place where the two differ (shown on a tube 0.10 m away)."""
    from ekf_slam_ml_amd import synth
    rng = np.random.default_rng(12)
    S = 300
    poses = np.stack([rng.uniform(-np.pi, np.pi, S), rng.uniform(-0.8, 0.8, S), rng.uniform(-0.8, 0.8, S)], axis=1)
    world = np.stack([synth.TUBE_X, synth.TUBE_Y], axis=1)
    near = np.sqrt(((poses[:, None, 1:] - world[None]) ** 2).sum(-1)).min(axis=1)
    far = near >= 0.145
    a = synth.make_scans(poses, world, range_std=0.0, model=0)
    b = synth.make_scans(poses, world, range_std=0.0, model=1)
    assert far.sum() > 200 and np.abs(a[far] - b[far]).max() < 1e-12
    assert (a[far] < 1.0).mean() > 0.2                       # walls and tubes are really hit
    # both models draw the same noise (same (seed, filter, step, beam) addressing)
    n1 = synth.make_scans(poses[far][:3], world, seed=5, model=1) - b[far][:3]
    n0 = synth.make_scans(poses[far][:3], world, seed=5, model=0) - a[far][:3]
    assert np.abs(n1 - n0).max() < 1e-12 and np.abs(n1).max() > 1e-3
    # a tube 0.10 m ahead: it subtends +-49.6 deg, the window is +-32.4 deg -> the reference sees the wall beyond its rim
    tube = np.array([[0.10, 0.0]])
    p0 = np.zeros((1, 3))
    c = synth.make_scans(p0, tube, range_std=0.0, model=0, border=10.0)[0]
    r = synth.make_scans(p0, tube, range_std=0.0, model=1, border=10.0)[0]
    assert abs(c[0] - (0.10 - synth.TUBE_RADIUS)) < 1e-12 and abs(r[0] - c[0]) < 1e-12      # straight ahead: both hit
    assert c[40] < 0.2 and r[40] == 3.5                                                    # 40 deg: inside the rim, outside the window
