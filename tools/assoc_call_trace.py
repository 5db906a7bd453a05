"""Phase trace of k_assoc_call (a whole data_association() call of a single filter in one launch) on configs[2]'s discovery
run: 100 MHz wall-clock stamps of thread 0, 16 slots per reading; microseconds since the kernel's first stamp, median over
the calls of the last 500 steps, per reading index."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth
T = 2000
log = synth.make_unknown_log(synth.config3(steps=T))
meas = [np.ascontiguousarray(log.meas_xy[t, 0, :log.count[t, 0]]) for t in range(T)]
f = capi.EKF_SLAM(1000)
k = np.zeros(1000, dtype=np.uint8)
names = ["loop top", "scored", "barrier 1", "decided (barrier 2)", "requests issued", "barrier 3", "gains stored", "barrier 4", "block folded"]
acc = {}
for t in range(T):
    if t == T - 500:
        f.phase_trace(True)
    f.prediction(tuple(log.twist[t, 0]))
    a = f.data_association(meas[t], k)
    if t >= T - 500 and len(a) > 0:
        tr = f.phase_trace(True, fetch=True).astype(np.float64).reshape(8, 16)
        t0 = tr[0, 15]
        if t0 == 0:
            continue
        for j in range(min(len(a), 8)):
            if a[j] >= 0:
                acc.setdefault(j, []).append((tr[j, :9] - t0) / 100.0)
f.phase_trace(False)
print("known landmarks at the end:", int(k.sum()))
for j in sorted(acc):
    m = np.median(np.stack(acc[j]), axis=0)
    print(f"reading {j} ({len(acc[j])} calls): " + "  ".join(f"{names[i]} {m[i]:.2f}" for i in range(9)))
f.close()
