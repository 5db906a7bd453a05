"""Extended differential check of pools' delayed data_association() (ekf_stepfused.hip, DELAYED + block cache) against the
eager run of the same pool: random pool sizes, map sizes, corrections per flush, surveyed shares, ragged reading counts,
filters that sit steps out, run boundaries.  Decisions and known counts must be identical, states within 1e-9."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (one HIP runtime)
from ekf_slam_ml_amd import capi as hip, synth

N_SCEN = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for seed in range(N_SCEN):
    rng = np.random.default_rng(90000 + seed)
    B, n, T = int(rng.integers(3, 20)), int(rng.integers(110, 340)), int(rng.integers(5, 15))
    k = int(rng.choice([8, 16, 24, 32, 40, 64]))
    share = float(rng.choice([1.0, 1.0, 0.7, 0.3]))
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=1000 + seed, half_extent=float(rng.uniform(4.0, 7.0)), min_spacing=0.3,
                          max_visible_dis=float(rng.uniform(1.0, 1.6)), vmax=8, v_cmd=1.0, w_cmd=0.6)
    log = synth.make_unknown_log(cfg)
    cnt = log.count.copy()
    cnt[rng.random(cnt.shape) < 0.08] = 0          # filters that sit a step out
    init = (log.world[None] + rng.normal(0.0, 0.005, size=(B, n, 2))).reshape(B, 2 * n)
    lm0 = np.full((2, B, 1), -1, dtype=np.int32)
    known0 = np.where(np.arange(B) < share * B, n, 0).astype(np.int32)
    cut = int(rng.integers(1, T))
    res = []
    for mode in (0, k):
        bt = hip.BatchEKF(B, n)
        bt.upload_known_log(np.zeros((2, B, 2)), lm0, np.zeros((2, B, 1, 2)), init)
        bt.run_known()
        bt.set_known_counts(known0)
        bt.set_update_mode(mode)
        bt.upload_unknown_log(log.twist, cnt, log.meas_xy)
        bt.run_unknown(0, cut)
        bt.run_unknown(cut, T)
        res.append((bt.decisions().copy(), bt.known_counts().copy(), np.stack([bt.state(b) for b in range(B)]), bt.cov(int(rng.integers(0, B)) if False else 0)))
        bt.close()
    ok = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    ds = float(np.abs(res[0][2] - res[1][2]).max())
    dc = float(np.abs(res[0][3] - res[1][3]).max() / np.abs(res[0][3]).max())
    if not ok or not (ds < 1e-9 and dc < 1e-9):
        bad += 1
        print(f"FAIL seed {seed}: B={B} n={n} T={T} k={k} share={share} decisions_equal={ok} dstate={ds:.2e} dcov={dc:.2e}", flush=True)
    if seed % 10 == 9:
        print("scenario", seed, "failures so far", bad, flush=True)
print("done, failures:", bad)
