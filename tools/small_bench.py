import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth
log = synth.make_known_log(synth.config1(steps=3000))
steps = [log.expand_step(t) for t in range(3000)]
tw = [tuple(log.twist[t, 0]) for t in range(3000)]
for small in (True, False):
    f = capi.EKF_SLAM(20); f.set_small_map_path(small)
    for t in range(100):
        f.prediction(tw[t]); f.measurement(*steps[t])
    f.sync(); t0 = time.perf_counter()
    for t in range(100, 3000):
        f.prediction(tw[t]); f.measurement(*steps[t])
    t1 = time.perf_counter(); f.sync(); t2 = time.perf_counter()
    print(f"small={small}: host loop {(t1 - t0) / 2900 * 1e6:.1f} us/step, incl. drain {(t2 - t0) / 2900 * 1e6:.1f} us/step")
    f.close()
