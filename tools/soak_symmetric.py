"""Extended differential check of the symmetric option of the delayed mode (row-only gain steps, mirrored flush, rows-only
prediction upkeep inside a run) against the eager run of the same pool: random pool sizes, map sizes (around the 256
threshold of the mirrored flush too), corrections per flush, reading slots per step, steps without any visible landmark,
run boundaries.  States within 1e-9, covariances within 1e-9 relative; after every run the covariance handed back must be
symmetric to the bit outside the 32 x 32 diagonal squares when the mirrored flush ran."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (one HIP runtime)
from ekf_slam_ml_amd import capi as hip, synth

N_SCEN = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for seed in range(N_SCEN):
    rng = np.random.default_rng(70000 + seed)
    B, n, T = int(rng.integers(1, 20)), int(rng.choice([40, 100, 126, 127, 128, 200, 333, 500, 700, 1000])), int(rng.integers(6, 30))
    k, vmax = int(rng.choice([1, 2, 3, 8, 16, 17, 32, 48, 64])), int(rng.choice([1, 2, 3, 5]))
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=3000 + seed, half_extent=float(rng.uniform(2.0, 6.0)) * max(1.0, (n / 150.0) ** 0.5), min_spacing=0.15,
                          max_visible_dis=1e9 if vmax == 1 else float(rng.uniform(0.8, 2.0)), vmax=vmax)
    log = synth.make_known_log(cfg)
    blind = rng.random(T) < 0.15
    log.lm_idx[blind] = -1
    cuts = sorted(set([0, T] + [int(c) for c in rng.integers(1, T, size=int(rng.integers(0, 3)))]))
    res = []
    for mode in (0, k):
        bt = hip.BatchEKF(B, n)
        bt.set_update_mode(mode, symmetric_gather=bool(mode))
        bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
        sym_ok = True
        for a, b in zip(cuts[:-1], cuts[1:]):
            bt.run_known(a, b)
            if mode and bt.form_counts()["flush_mirrored"]:
                c = bt.cov(int(rng.integers(0, B)))
                tile = np.arange(c.shape[0]) // 32
                above = tile[:, None] < tile[None, :]
                sym_ok = sym_ok and np.array_equal(c.T[above], c[above])
        res.append((np.stack([bt.state(b) for b in range(B)]), bt.cov(0), bt.cov(B - 1), sym_ok))
        bt.close()
    ds = float(np.abs(res[0][0] - res[1][0]).max())
    dc = max(float(np.abs(res[0][i] - res[1][i]).max() / np.abs(res[0][i]).max()) for i in (1, 2))
    if not (ds < 1e-9 and dc < 1e-9 and res[1][3]):
        bad += 1
        print(f"FAIL seed {seed}: B={B} n={n} T={T} k={k} vmax={vmax} cuts={cuts} dstate={ds:.2e} dcov={dc:.2e} mirror={res[1][3]}", flush=True)
    if seed % 10 == 9:
        print("scenario", seed, "failures so far", bad, flush=True)
print("done, failures:", bad)
