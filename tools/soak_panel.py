"""Extended differential check of the column panel of delayed known-association runs (EKF_FORM_COLUMN_PANEL): random pools
(1-19 filters, n = 100 ... 1000 around the panel's minimum dimension 256, 1-64 corrections per flush, 1-6 reading slots per
step, blind steps, filters that sit steps out, random run boundaries with and without a getter in between, plain and strip
flush) with the panel on, with the one-slot plan, and with the panel off: state and covariance of every filter must be
BIT-identical between the three; with the kept current rows / columns on top (the default) within 1e-10 of them; and all
within 1e-9 of the eager run.   python tools/soak_panel.py [N=60] [first_seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ekf_slam_ml_amd import capi as hip, synth

N_SCEN = int(sys.argv[1]) if len(sys.argv) > 1 else 60
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(FIRST, FIRST + N_SCEN):
    rng = np.random.default_rng(81000 + seed)
    B, n, T = int(rng.integers(1, 20)), int(rng.choice([100, 126, 127, 128, 200, 333, 500, 1000])), int(rng.integers(6, 30))
    k, vmax = int(rng.choice([1, 2, 3, 4, 8, 16, 17, 32, 64])), int(rng.choice([1, 2, 2, 3, 5, 6]))
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=5000 + seed, half_extent=float(rng.uniform(2.0, 6.0)) * max(1.0, (n / 150.0) ** 0.5),
                          min_spacing=0.15, max_visible_dis=1e9 if vmax <= 2 else float(rng.uniform(0.8, 2.0)), vmax=vmax,
                          v_cmd=float(rng.uniform(0.1, 1.0)), w_cmd=float(rng.uniform(0.05, 0.6)))
    log = synth.make_known_log(cfg)
    lm = log.lm_idx.copy()
    lm[rng.random(T) < 0.12] = -1                                  # blind steps (the repair path when a run ends on them)
    for _ in range(int(rng.integers(0, 4))):
        lm[int(rng.integers(0, T)), int(rng.integers(0, B))] = -1  # a filter sits a step out
    cuts = sorted(set([0, T] + [int(c) for c in rng.integers(1, T, size=int(rng.integers(0, 4)))]))
    getter = rng.random(len(cuts)) < 0.3
    strip = bool(rng.integers(0, 2))
    base = hip.FORMS_DEFAULT | (hip.FORM_STRIP_FLUSH_ALWAYS if strip else 0)
    outs = {}
    nocur = base & ~hip.FORM_CURRENT_COLUMNS
    for name, forms, mode in (("panel", nocur, k), ("one_slot", nocur | hip.FORM_COLUMN_PANEL_ONE_SLOT, k),
                              ("off", base & ~hip.FORM_COLUMN_PANEL, k), ("current", base, k), ("eager", base, 0)):
        bt = hip.BatchEKF(B, n)
        bt.set_forms(forms)
        bt.set_update_mode(mode)
        bt.upload_known_log(log.twist, lm, log.z_xy, log.init_xy)
        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            bt.run_known(a, b)
            if getter[i]:
                bt.state(0)                                        # (drops the panel: the next run starts without one)
        outs[name] = (np.stack([bt.state(b) for b in range(B)]), [bt.cov(b) for b in sorted({0, B // 2, B - 1})], bt.form_counts())
        bt.close()
    same = all(np.array_equal(outs[v][0], outs["off"][0]) and all(np.array_equal(x, y) for x, y in zip(outs[v][1], outs["off"][1]))
               for v in ("panel", "one_slot"))
    ds = max(float(np.abs(outs[v][0] - outs["eager"][0]).max()) for v in ("panel", "current"))
    dc = max(float(np.abs(x - y).max() / np.abs(y).max()) for v in ("panel", "current") for x, y in zip(outs[v][1], outs["eager"][1]))
    # the kept current rows / columns (the default): another association of the same sums -- 1e-10 from the rebuilt form
    dcur = max(float(np.abs(outs["current"][0] - outs["off"][0]).max()),
               max(float(np.abs(x - y).max() / np.abs(y).max()) for x, y in zip(outs["current"][1], outs["off"][1])))
    same = same and dcur < 1e-10
    if not (same and ds < 1e-9 and dc < 1e-9):
        bad += 1
        print(f"FAIL seed {seed}: B={B} n={n} T={T} k={k} vmax={vmax} cuts={cuts} strip={strip} identical={same} dstate={ds:.2e} dcov={dc:.2e}", flush=True)
    if (seed - FIRST) % 10 == 9:
        print("scenario", seed, "failures so far", bad, "| last: panel launches", outs["panel"][2]["gain_from_panel"], flush=True)
print("done, failures:", bad)
