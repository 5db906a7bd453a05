"""CPU sanitizer leg ("ASan/UBSan on our host code + CPU restatement"; the reference has none,
rigid2d/CMakeLists.txt:128-135).  Never on the GPU build -- GPU AddressSanitizer is not available on this pool:
  * oracle/ekf_oracle.c + circle_oracle.c built with gcc -fsanitize=address,undefined (`make -C oracle asan`), and the
    restatement checks of tests/test_oracle.py / tests/test_circle_oracle.py re-run on that build in a libasan-preloaded
    child interpreter (golden vectors, dense vs structured vs NumPy, margins, OpenMP batch replay);
  * the ROS-free node loop tests/cpp/slam_replay.cpp over the C++ host mirror (ekf_slam_ml_amd/host/ekf_slam.hpp) built
    the same way: log parsing, marshalling, rule-of-five plumbing and the error path (no device -> exception -> exit 1)
    run without a GPU.
Zero sanitizer reports allowed."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.fixture(scope="module")
def asan_build():
    if _libasan() is None:
        pytest.skip("gcc's libasan.so is not installed")
    if not os.path.exists(os.path.join(ROOT, "ekf_slam_ml_amd", "libekfslam_hip.so")):
        pytest.fail("libekfslam_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], check=True)
    return _libasan()


def _env(libasan, **extra):
    env = dict(os.environ)
    env.update(LD_PRELOAD=libasan, EKF_ORACLE_SANITIZED="1",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",   # (CPython itself leaks by design)
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", **extra)
    return env


def _clean(text):
    return not any(m in text for m in ("ERROR: AddressSanitizer", "runtime error:", "ERROR: LeakSanitizer"))


def test_restatement_checks_under_asan_ubsan(asan_build):
    env = _env(asan_build)
    # the child really runs on the sanitized build
    r = subprocess.run([sys.executable, "-c", "from oracle import binding as b; print(b.lib()._name)"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("libekf_oracle_asan.so"), r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle.py",
                        "tests/test_circle_oracle.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and _clean(out), out[-4000:]


def test_host_mirror_node_loop_under_asan_ubsan(asan_build, tmp_path):
    exe = os.path.join(ROOT, "oracle", "slam_replay_asan")
    log = tmp_path / "log.txt"
    # header: mode n T wheel_base wheel_radius; then per step: dl dr count, count markers (id x y add), hex floats
    lines = ["0 6 3 0.16 0.033"]
    for t in range(3):
        lines.append(f"{(0.01 * (t + 1)).hex()} {(0.012 * (t + 1)).hex()} 6")
        lines += [f"{i} {(0.3 + 0.1 * i).hex()} {(0.2 - 0.05 * i).hex()} {i % 2}" for i in range(6)]
    log.write_text("\n".join(lines) + "\n")
    env = _env(asan_build)
    env.pop("LD_PRELOAD")   # (the executable links libasan itself)
    r = subprocess.run([exe, str(log), str(tmp_path / "out.txt")], env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert _clean(out), out[-4000:]
    # with a gfx950 device the replay completes (0); without one the constructor's EKF_ERR_NO_DEVICE becomes the mirror's
    # exception and the driver's exit code 1 -- there is no CPU path to fall back to
    assert r.returncode in (0, 1), out[-2000:]
    if r.returncode == 1:
        assert "no HIP device" in out or "device" in out.lower(), out[-2000:]
    # a malformed log is refused by the parser, cleanly
    bad = tmp_path / "bad.txt"
    bad.write_text("0 6 3 0.16\n")
    r = subprocess.run([exe, str(bad), str(tmp_path / "out2.txt")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and _clean(r.stdout + r.stderr)
