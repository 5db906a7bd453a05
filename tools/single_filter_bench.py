"""Single-filter timings on the GPU box: configs[1] (n=200 known) and configs[2] (n=1000 unknown)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth


def known(n, cfg, n_steps, modes=(-1, 0, 8, 16, 32)):
    log = synth.make_known_log(cfg)
    steps = [log.expand_step(t) for t in range(n_steps)]
    for k in modes:  # -1: eager, two-launch form (call-fused forms off)
        f = capi.EKF_SLAM(n)
        if k < 0:
            f.set_call_fused(False)
        f.set_update_mode(max(k, 0))
        for t in range(20):
            f.prediction(log.twist[t, 0]); f.measurement(*steps[t])
        f.sync()
        t0 = time.perf_counter()
        for t in range(20, n_steps):
            f.prediction(log.twist[t, 0]); f.measurement(*steps[t])
        f.sync()
        dt = time.perf_counter() - t0
        corr = int((log.lm_idx[20:] >= 0).sum())
        N = 3 + 2 * n
        print(f"n={n} known, update mode k={k:2d}: {(n_steps - 20) / dt:7.0f} steps/s, {corr / dt:8.0f} corrections/s, "
              f"{dt / corr * 1e6:6.1f} us/correction (V~{corr / (n_steps - 20):.1f}); eager-equivalent alg GB/s {corr * 16 * N * N / dt / 1e9:.0f}", flush=True)
        f.close()


def unknown(n_steps=150, fused=True):
    log = synth.make_unknown_log(synth.config3(steps=n_steps))
    f = capi.EKF_SLAM(1000)
    f.set_call_fused(fused)
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
    print(f"configs[2] n=1000 unknown (fused={fused}): {(n_steps - 20) / dt:.0f} steps/s, {meas / dt:.0f} measurements/s, {upd / dt:.0f} corrections/s, "
          f"{dt / max(meas, 1) * 1e6:.1f} us/measurement, known={int(k.sum())}", flush=True)
    f.close()


known(200, synth.config2(steps=400), 400)
known(1000, synth.config3(steps=120), 120)
unknown(fused=False)
unknown(fused=True)
