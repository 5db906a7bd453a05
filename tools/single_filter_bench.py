"""Single-filter timings on the GPU box: configs[1] (n=200 known) and configs[2] (n=1000 unknown)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth

def known(n_steps=400):
    log = synth.make_known_log(synth.config2(steps=n_steps))
    f = capi.EKF_SLAM(200)
    steps = [log.expand_step(t) for t in range(n_steps)]
    for t in range(50):
        f.prediction(log.twist[t, 0]); f.measurement(*steps[t])
    f.sync()
    t0 = time.perf_counter()
    for t in range(50, n_steps):
        f.prediction(log.twist[t, 0]); f.measurement(*steps[t])
    f.sync()
    dt = time.perf_counter() - t0
    corr = int((log.lm_idx[50:] >= 0).sum())
    print(f"configs[1] n=200 known: {(n_steps-50)/dt:.0f} steps/s, {corr/dt:.0f} corrections/s, {dt/corr*1e6:.1f} us/correction "
          f"(V~{corr/(n_steps-50):.1f}); alg GB/s {corr*16*403**2/dt/1e9:.1f}", flush=True)
    f.close()

def unknown(n_steps=150):
    log = synth.make_unknown_log(synth.config3(steps=n_steps))
    f = capi.EKF_SLAM(1000)
    k = np.zeros(1000, dtype=np.uint8)
    for t in range(20):
        f.prediction(log.twist[t, 0]); f.data_association(log.meas_xy[t, 0, :log.count[t, 0]], k)
    f.sync()
    t0 = time.perf_counter()
    upd = 0; meas = 0
    for t in range(20, n_steps):
        f.prediction(log.twist[t, 0])
        a = f.data_association(log.meas_xy[t, 0, :log.count[t, 0]], k)
        upd += int((a >= 0).sum()); meas += len(a)
    f.sync()
    dt = time.perf_counter() - t0
    print(f"configs[2] n=1000 unknown: {(n_steps-20)/dt:.0f} steps/s, {meas/dt:.0f} measurements/s, {upd/dt:.0f} corrections/s, "
          f"{dt/max(upd,1)*1e6:.1f} us/correction, known={int(k.sum())}; alg GB/s {upd*16*2003**2/dt/1e9:.1f}", flush=True)
    f.close()

def known1000(n_steps=120):
    cfg = synth.config3(steps=n_steps); 
    log = synth.make_known_log(cfg)
    f = capi.EKF_SLAM(1000)
    steps = [log.expand_step(t) for t in range(n_steps)]
    for t in range(20):
        f.prediction(log.twist[t, 0]); f.measurement(*steps[t])
    f.sync()
    for rows, u in ((0, 0), (4, 4), (8, 8), (16, 8), (16, 16), (32, 16)):
        f.set_tuning(rows, -1, u)
        t0 = time.perf_counter()
        for t in range(20, n_steps):
            f.prediction(log.twist[t, 0]); f.measurement(*steps[t])
        f.sync()
        dt = time.perf_counter() - t0
        corr = int((log.lm_idx[20:] >= 0).sum())
        print(f"n=1000 known single filter rows={rows} U={u}: {corr/dt:.0f} corrections/s, {dt/corr*1e6:.1f} us/correction; alg GB/s {corr*16*2003**2/dt/1e9:.0f}", flush=True)
    f.close()

known(); unknown(); known1000()
