"""GPU box: data_association() of one n = 1000 filter on a fully discovered map (configs[2] full-map leg of bench.py),
call-fused vs round-1 path; for rocprofv3 kernel traces."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth

n, steps = 1000, 200
cfg = synth.config3(steps=steps)
log = synth.make_unknown_log(cfg)
meas = [log.meas_xy[t, 0, :log.count[t, 0]] for t in range(steps)]
variants = [(True, 0, -1, 0), (False, 0, -1, 0)]
if len(sys.argv) > 1:   # tuning sweep of the once-per-call pass: rows,nt,group ...
    variants = [(True,) + tuple(int(x) for x in v.split(",")) for v in sys.argv[1:]]
for cf, rows, nt, group in variants:
    rng = np.random.default_rng(33)
    f = capi.EKF_SLAM(n)
    f.set_call_fused(cf)
    f.set_tuning(rows, nt, group)
    rel = synth._robot_frame(log.world, np.zeros((1, 3)))[0]
    f.measurement((rel + rng.normal(0, 0.005, rel.shape)).reshape(-1), np.zeros(n, dtype=np.uint8))
    f.measurement((rel + rng.normal(0, 0.005, rel.shape)).reshape(-1), np.ones(n, dtype=np.uint8))
    kn = np.ones(n, dtype=np.uint8)
    f.sync()
    t0 = time.perf_counter()
    nm = 0
    for t in range(steps):
        f.prediction(log.twist[t, 0]); nm += len(f.data_association(meas[t], kn))
    f.sync()
    dt = time.perf_counter() - t0
    print(f"call_fused={cf} rows={rows} nt={nt} group={group}: {nm / dt:8.0f} measurements/s, {dt / nm * 1e6:6.1f} us per measurement, {dt / steps * 1e6:7.1f} us per call", flush=True)
    f.close()
