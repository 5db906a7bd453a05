"""configs[2] single filter for a few hundred steps, repeated (run under rocprofv3 --kernel-trace to see per-kernel time
and gaps; the repeats show the run-to-run spread on one box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 400
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
log = synth.make_unknown_log(synth.config3(steps=T))
meas = [np.ascontiguousarray(log.meas_xy[t, 0, :log.count[t, 0]]) for t in range(T)]
tw = [tuple(log.twist[t, 0]) for t in range(T)]
for rep in range(reps):
    f = capi.EKF_SLAM(1000)
    k = np.zeros(1000, dtype=np.uint8)
    t0 = time.perf_counter(); nm = 0
    for t in range(T):
        f.prediction(tw[t])
        a = f.data_association(meas[t], k); nm += len(a)
    f.sync()
    dt = time.perf_counter() - t0
    print(f"{T / dt:.0f} steps/s, {dt / nm * 1e6:.1f} us/measurement, {dt / T * 1e6:.1f} us/step, known={int(k.sum())}", flush=True)
    f.close()
