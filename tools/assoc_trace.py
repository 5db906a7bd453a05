"""Phase trace of k_assoc_reading (one launch per reading of a call-fused data_association()) on the full map of
configs[2]: n = 1000, every landmark known.  100 MHz wall-clock stamps of thread 0 of workgroup 0, 16 slots per reading;
printed as microseconds since the launch's first stamp, median over calls, per position pc of the reading in its pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth

n, steps = 1000, 150
cfg = synth.config3(steps=steps)
log = synth.make_unknown_log(cfg)
meas = [log.meas_xy[t, 0, :log.count[t, 0]] for t in range(steps)]
rng = np.random.default_rng(33)
f = capi.EKF_SLAM(n)
rel = synth._robot_frame(log.world, np.zeros((1, 3)))[0]
f.measurement((rel + rng.normal(0, 0.005, rel.shape)).reshape(-1), np.zeros(n, dtype=np.uint8))
f.measurement((rel + rng.normal(0, 0.005, rel.shape)).reshape(-1), np.ones(n, dtype=np.uint8))
kn = np.ones(n, dtype=np.uint8)
f.phase_trace(True)
acc = {}
names = ["start", "requests issued / scanned", "barrier 1", "decided (barrier 2)", "gathers issued", "block - pending pairs",
         "barrier 3", "gain done", "published (barrier 4)", "own pair folded", "predicted_terms", "end"]
for t in range(steps):
    f.prediction(log.twist[t, 0]); f.data_association(meas[t], kn)
    tr = f.phase_trace(True, fetch=True).astype(np.float64).reshape(8, 16)
    J = len(meas[t])
    for pc in range(min(J, 8)):
        row = (tr[pc, :12] - tr[pc, 0]) / 100.0
        acc.setdefault((pc, pc + 1 < J), []).append(row)
f.phase_trace(False)
for key in sorted(acc):
    a = np.median(np.stack(acc[key]), axis=0)
    last = 12 if key[1] else 8
    print(f"pc = {key[0]} next reading: {key[1]} ({len(acc[key])} launches): " + "  ".join(f"{names[k]} {a[k]:.2f}" for k in range(1, last)))
f.close()
