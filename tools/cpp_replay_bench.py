"""The C++ node loop (tests/cpp/slam_replay: Odometer + callbacks + main_loop over the header-only mirror class)
timed on BASELINE configs[0..2] shapes: what a compiled caller pays per step, without Python in the loop."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ekf_slam_ml_amd import synth
BIN = os.path.join(ROOT, "tests", "cpp", "slam_replay")
hx = lambda v: float(v).hex()


def known(cfg, tag):
    log = synth.make_known_log(cfg)
    n, T = cfg.n, cfg.steps
    lines = [f"0 {n} {T} {hx(synth.WHEEL_BASE)} {hx(synth.WHEEL_RADIUS)}"]
    for t in range(T):
        sensor, vis = log.expand_step(t)
        lines.append(f"{hx(log.wheel[t, 0, 0])} {hx(log.wheel[t, 0, 1])} {n}")
        lines += [f"{i} {hx(sensor[2 * i])} {hx(sensor[2 * i + 1])} {int(vis[i]) if t else 1}" for i in range(n)]
    run(lines, tag)


def unknown(cfg, tag):
    log = synth.make_unknown_log(cfg)
    n, T = cfg.n, cfg.steps
    lines = [f"1 {n} {T} {hx(synth.WHEEL_BASE)} {hx(synth.WHEEL_RADIUS)}"]
    for t in range(T):
        J = int(log.count[t, 0])
        lines.append(f"{hx(log.wheel[t, 0, 0])} {hx(log.wheel[t, 0, 1])} {J}")
        lines += [f"{j} {hx(log.meas_xy[t, 0, j, 0])} {hx(log.meas_xy[t, 0, j, 1])} 1" for j in range(J)]
    run(lines, tag)


def run(lines, tag):
    with tempfile.TemporaryDirectory() as d:
        lp, op = os.path.join(d, "log.txt"), os.path.join(d, "out.txt")
        open(lp, "w").write("\n".join(lines) + "\n")
        r = subprocess.run([BIN, lp, op], capture_output=True, text=True, timeout=600)
        print(tag, "|", r.stderr.strip().splitlines()[-1] if r.stderr.strip() else f"rc={r.returncode}", flush=True)


known(synth.config1(steps=3000), "configs[0] n=20 known")
known(synth.config2(steps=2000), "configs[1] n=200 known")
unknown(synth.config3(steps=1000), "configs[2] n=1000 unknown")
