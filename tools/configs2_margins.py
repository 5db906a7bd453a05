"""Decision margins of bench.py's configs_2 legs on the CPU checker (no GPU needed): for the discovery run and for the
full-map phase (map built through the known-association API from `seed`), the relative distance of every score to the
gates 10.0 / 1.0 (ekf_slam.cpp:293,330), the winner-to-runner-up gap (:305-309) and the decision-relevant minimum.
    python tools/configs2_margins.py [seed ...]          (default: bench.py's FULL_MAP_SEED)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ekf_slam_ml_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402  (checker: this is a test tool)


def full_map_margins(seed, steps=300):
    cfg = synth.config3(steps=steps)
    log = synth.make_unknown_log(cfg)
    n = cfg.n
    rng = np.random.default_rng(seed)

    def all_readings():
        rel = synth._robot_frame(log.world, np.zeros((1, 3)))[0]
        return (rel + rng.normal(0.0, cfg.sensor_std, size=rel.shape)).reshape(-1)

    o = ob.OracleEKF(n, ob.STRUCTURED, fast=True)
    o.measurement(all_readings(), np.zeros(n, dtype=np.uint8))
    o.measurement(all_readings(), np.ones(n, dtype=np.uint8))
    kn = np.ones(n, dtype=np.uint8)
    m = ob.new_margins()
    for t in range(steps):
        o.prediction(*log.twist[t, 0])
        o.data_association(log.meas_xy[t, 0, :log.count[t, 0]], kn, m)
    return m


if __name__ == "__main__":
    import bench
    seeds = [int(x) for x in sys.argv[1:]] or [bench.FULL_MAP_SEED]
    for sd in seeds:
        m = full_map_margins(sd)
        print(sd, {k: float(f"{v:.4g}") for k, v in zip(ob.MARGIN_KEYS, m)}, "min of the first three:", float(f"{m[:3].min():.4g}"))
