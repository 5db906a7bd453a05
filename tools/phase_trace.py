"""Phase trace of k_call_factors (the factor kernel of the two-launch measurement() call) on configs[1]: a single filter,
n = 200, known association.  Stamps are 100 MHz wall-clock ticks of lane 0 of the control wave (row 0) and of the first
slice wave (row 1) of workgroup 0; printed as microseconds since the kernel's first stamp, averaged per call size V.
  python tools/phase_trace.py [n] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
cfg = synth.config2(steps=steps)
cfg.n = n
log = synth.make_known_log(cfg)
inputs = [log.expand_step(t) for t in range(steps)]
f = capi.EKF_SLAM(n)
for t in range(60):
    f.prediction(log.twist[t, 0]); f.measurement(*inputs[t])
f.phase_trace(True)
by_v = {}
clk = []
for t in range(60, steps):
    V = int(inputs[t][1].sum())
    f.prediction(log.twist[t, 0]); f.measurement(*inputs[t])
    tr = f.phase_trace(True, fetch=True).astype(np.float64)
    if V == 0 or V > 8 or tr[0, 0] == 0:
        continue
    clk.append((tr[0, 62] - tr[0, 61]) / max(tr[0, 60] - tr[0, 0], 1.0) * 100.0)   # shader cycles per 10-ns tick -> MHz
    by_v.setdefault(V, []).append((tr - tr[0, 0]) / 100.0)   # us
f.phase_trace(False)
for V in sorted(by_v):
    a = np.median(np.stack(by_v[V]), axis=0)
    print(f"V = {V} ({len(by_v[V])} calls): control wave: gathers issued {a[0,1]:.2f}  prologue done {a[0,2]:.2f}  end {a[0,60]:.2f} us"
          f" | slice wave: end {a[1,60]:.2f}")
    for t in range(V):
        c = a[0, 3 + 5 * t: 7 + 5 * t]
        s = a[1, 3 + 5 * t: 5 + 5 * t]
        print(f"   correction {t}: barrier A {c[0]:.2f}  terms_h(t+1) done {c[1]:.2f}  barrier B {c[2]:.2f}  S, S^-1, gains(t+1) done {c[3]:.2f}"
              f" | slice: barrier A {s[0]:.2f}  panels done {s[1]:.2f}")
print(f"in-kernel shader clock (median over calls): {np.median(clk):.0f} MHz")
# wall time per tick with the trace off
f.sync()
t0 = time.perf_counter()
for t in range(60, steps):
    f.prediction(log.twist[t, 0]); f.measurement(*inputs[t])
f.sync()
dt = time.perf_counter() - t0
corr = int(sum(inputs[t][1].sum() for t in range(60, steps)))
print(f"n = {n}: {dt / (steps - 60) * 1e6:.2f} us per tick, {corr / dt:.0f} corrections/s ({corr / (steps - 60):.2f} per tick)")
f.close()
