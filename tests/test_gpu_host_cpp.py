"""-m gpu: the C++ mirror class (ekf_slam_ml_amd/host/ekf_slam.hpp) driven by a ROS-free replay of the
nuslam node loop (tests/cpp/slam_replay.cpp: Odometer::getCurrentTwist, callback_fake_sensor /
callback_scan_sensor, main_loop INIT/UPDATE sequencing, copy-assignment of the filter object) -- the
"drops into the existing node" claim, checked against the CPU checker."""
import os
import subprocess

import numpy as np
import pytest

from ekf_slam_ml_amd import synth
from parity import FP64_TOL, assert_parity

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "cpp", "slam_replay")


def _hex(v):
    return float(v).hex()


def _run(tmp_path, lines, n):
    if not os.path.exists(BIN):
        subprocess.run(["make", "-C", os.path.join(HERE, "cpp"), "-s"], check=True)
    log, out = tmp_path / "log.txt", tmp_path / "out.txt"
    log.write_text("\n".join(lines) + "\n")
    subprocess.run([BIN, str(log), str(out)], check=True, timeout=300)
    vals = out.read_text().split()
    N = int(vals[0])
    assert N == 3 + 2 * n
    nums = [float.fromhex(v) if v.startswith(("0x", "-0x")) else float(v) for v in vals[1:]]
    state = np.array(nums[:N])
    cov = np.array(nums[N:N + N * N]).reshape(N, N)
    known = np.array(nums[N + N * N:N + N * N + n], dtype=np.uint8)
    tail = nums[N + N * N + n:]
    return state, cov, known, tail


def test_known_association_node_loop(hip, oracle, tmp_path):
    n, T = 20, 60
    log = synth.make_known_log(synth.config1(steps=T))
    lines = [f"0 {n} {T} {_hex(synth.WHEEL_BASE)} {_hex(synth.WHEEL_RADIUS)}"]
    o = oracle.OracleEKF(n, oracle.DENSE)
    for t in range(T):
        sensor, vis = log.expand_step(t)
        lines.append(f"{_hex(log.wheel[t, 0, 0])} {_hex(log.wheel[t, 0, 1])} {n}")
        for i in range(n):
            lines.append(f"{i} {_hex(sensor[2 * i])} {_hex(sensor[2 * i + 1])} {int(vis[i]) if t else 1}")
        tw = synth.body_twist(log.wheel[t, 0, 0] * 10.0, log.wheel[t, 0, 1] * 10.0)
        assert tw[0] == log.twist[t, 0, 0] and tw[1] == log.twist[t, 0, 1]
        o.prediction(*tw)
        o.measurement(sensor, vis)
    state, cov, known, tail = _run(tmp_path, lines, n)
    assert_parity(state, cov, o.state, o.cov, FP64_TOL, "C++ node loop, known association")
    seen = np.zeros(n, dtype=np.uint8)
    for t in range(1, T):
        seen |= log.expand_step(t)[1]
    assert np.array_equal(known, seen)  # known_list bookkeeping of callback_fake_sensor (slam.cpp:320-322)
    assert abs(tail[0] - o.state[0]) < 1e-9 and abs(tail[1] - o.state[1]) < 1e-9 and abs(tail[3] - o.state[-1]) < 1e-9


def test_unknown_association_node_loop(hip, oracle, tmp_path):
    n, T = 20, 50
    cfg = synth.config1(steps=T)
    cfg.seed = 5150
    log = synth.make_unknown_log(cfg)
    lines = [f"1 {n} {T} {_hex(synth.WHEEL_BASE)} {_hex(synth.WHEEL_RADIUS)}"]
    o = oracle.OracleEKF(n, oracle.DENSE)
    known = np.zeros(n, dtype=np.uint8)
    for t in range(T):
        J = int(log.count[t, 0])
        lines.append(f"{_hex(log.wheel[t, 0, 0])} {_hex(log.wheel[t, 0, 1])} {J}")
        for j in range(J):
            lines.append(f"{j} {_hex(log.meas_xy[t, 0, j, 0])} {_hex(log.meas_xy[t, 0, j, 1])} 1")
        o.prediction(*log.twist[t, 0])
        o.data_association(log.meas_xy[t, 0, :J], known)
    state, cov, known_cpp, _ = _run(tmp_path, lines, n)
    assert np.array_equal(known_cpp, known) and known.sum() >= 5
    assert_parity(state, cov, o.state, o.cov, FP64_TOL, "C++ node loop, unknown association")


def test_scan_pipeline_node_loop(hip, oracle, tmp_path):
    """landmarks node + unknown_data_assoc node in C++: laser ranges -> ekfslam::CircleFitting ->
    EKF_SLAM::data_association, against the checker's circle fitting + filter."""
    n, T = 10, 40
    cfg = synth.config1(steps=T)
    cfg.seed = 4711
    log = synth.make_unknown_log(cfg)
    world = np.stack([synth.TUBE_X, synth.TUBE_Y], axis=1)
    scans = synth.make_scans(log.true_pose[:, 0], world=world, seed=12)
    lines = [f"2 {n} {T} {_hex(synth.WHEEL_BASE)} {_hex(synth.WHEEL_RADIUS)}"]
    o = oracle.OracleEKF(n, oracle.DENSE)
    known = np.zeros(n, dtype=np.uint8)
    for t in range(T):
        lines.append(f"{_hex(log.wheel[t, 0, 0])} {_hex(log.wheel[t, 0, 1])} {scans.shape[1]}")
        for i, r in enumerate(scans[t]):
            lines.append(f"{i} {_hex(r)} {_hex(0.0)} 1")
        circles, _, _ = oracle.approx_circle_positions(scans[t])
        o.prediction(*log.twist[t, 0])
        o.data_association(circles, known)
    state, cov, known_cpp, _ = _run(tmp_path, lines, n)
    assert np.array_equal(known_cpp, known) and known.sum() >= 3
    assert_parity(state, cov, o.state, o.cov, FP64_TOL, "C++ scan -> circles -> association")
