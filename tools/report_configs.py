"""BASELINE.json configs[0..3] on the GPU box: timing + parity figures for the table in DESIGN.md section 5.
(configs[4] is bench.py.)  Prints one line per configuration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ekf_slam_ml_amd import capi, synth
from oracle import binding as ob      # checker: parity figures and CPU timings only
from parity import worst


def known(tag, cfg, oracle_mode, oracle_steps, k_modes=(0, 16)):
    log = synth.make_known_log(cfg)
    n, T = cfg.n, cfg.steps
    steps = [log.expand_step(t) for t in range(T)]
    o = ob.OracleEKF(n, oracle_mode)
    t0 = time.perf_counter()
    for t in range(oracle_steps):
        o.prediction(*log.twist[t, 0]); o.measurement(*steps[t])
    cpu_dt = time.perf_counter() - t0
    cpu_corr = int((log.lm_idx[:oracle_steps] >= 0).sum())
    for k in k_modes:
        f = capi.EKF_SLAM(n)
        f.set_update_mode(k)
        for t in range(oracle_steps):
            f.prediction(log.twist[t, 0]); f.measurement(*steps[t])
        w, _ = worst(f.state, f.cov, o.state, o.cov)
        f.sync()
        t0 = time.perf_counter()
        for t in range(oracle_steps, T):
            f.prediction(log.twist[t, 0]); f.measurement(*steps[t])
        f.sync()
        dt = time.perf_counter() - t0
        corr = int((log.lm_idx[oracle_steps:] >= 0).sum())
        N = 3 + 2 * n
        print(f"{tag} [{'eager' if k == 0 else f'delayed k={k}'}]: {(T - oracle_steps) / dt:8.0f} steps/s  {corr / dt:8.0f} corrections/s "
              f"(V~{corr / (T - oracle_steps):.1f})  eager-equivalent {corr * 16 * N * N / dt / 1e9:7.1f} GB/s  "
              f"parity vs {'dense' if oracle_mode == ob.DENSE else 'structured'} checker after {oracle_steps} steps: {w:.1e}  "
              f"| CPU checker ({'dense O(N^3)' if oracle_mode == ob.DENSE else 'structured'}, 1 thread): {cpu_corr / cpu_dt:.0f} corrections/s", flush=True)
        f.close()


def unknown(tag, cfg, oracle_steps):
    log = synth.make_unknown_log(cfg)
    n, T = cfg.n, cfg.steps
    f, o = capi.EKF_SLAM(n), ob.OracleEKF(n, ob.STRUCTURED)
    kf, ko = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    same = True
    t0 = time.perf_counter()
    for t in range(oracle_steps):
        m = log.meas_xy[t, 0, :log.count[t, 0]]
        o.prediction(*log.twist[t, 0]); b = o.data_association(m, ko)
    cpu_dt = time.perf_counter() - t0
    for t in range(oracle_steps):
        m = log.meas_xy[t, 0, :log.count[t, 0]]
        f.prediction(log.twist[t, 0]); a = f.data_association(m, kf)
    same = np.array_equal(kf, ko)
    w, _ = worst(f.state, f.cov, o.state, o.cov)
    t0 = time.perf_counter()
    meas = upd = 0
    for t in range(oracle_steps, T):
        m = log.meas_xy[t, 0, :log.count[t, 0]]
        f.prediction(log.twist[t, 0]); a = f.data_association(m, kf)
        meas += len(a); upd += int((a >= 0).sum())
    f.sync()
    dt = time.perf_counter() - t0
    print(f"{tag}: {(T - oracle_steps) / dt:8.0f} steps/s  {meas / dt:8.0f} measurements/s (J~{meas / (T - oracle_steps):.1f}, "
          f"M grows to {int(kf.sum())})  {upd / dt:8.0f} corrections/s  known_list identical: {same}  parity after {oracle_steps} steps: {w:.1e}  "
          f"| CPU structured checker: {int(log.count[:oracle_steps].sum()) / cpu_dt:.0f} measurements/s", flush=True)
    f.close()


known("configs[0] n=20 known, 1000 steps", synth.config1(steps=1000), ob.DENSE, 300)
known("configs[1] n=200 known, 2000 steps", synth.config2(steps=2000), ob.DENSE, 12)
unknown("configs[2] n=1000 unknown, 2000 steps", synth.config3(steps=2000), 40)
