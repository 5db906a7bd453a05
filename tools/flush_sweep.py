"""Flush time of the delayed update at BASELINE.json configs[4]'s shape: the plain form (k_flush), the strip form
(k_flush_strip; automatic beyond 40 pending vectors on pools that fill the chip) and the mirrored form of the symmetric
option (k_flush_sym), non-temporal access on / off, k = corrections per flush (two pending vectors each).
usage: python tools/flush_sweep.py [B=4096] [k,k,...] [nt,nt] [forms=plain,strip,mirrored] [n=1000]
(rocprofv3 --pmc FETCH_SIZE -- python3 tools/flush_sweep.py 4096 32 1   collects the HBM read traffic of the forms)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekf_slam_ml_amd import capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
KS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 8, 16, 32]
NTS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 0]
FORMS = sys.argv[4].split(",") if len(sys.argv) > 4 else ["plain", "strip", "mirrored"]
NL = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
W = 2
K = max(KS)
cfg = synth.config5(filters=B, steps=1 + W + K, n=NL)
bt = capi.BatchEKF(B, NL)
bt.simulate_known_log(cfg, synth.make_world(NL, cfg.half_extent, cfg.min_spacing, cfg.world_seed))
for k in KS:
    for name in FORMS:
        if name == "strip" and 2 * k > 80:
            continue
        for nt in NTS:
            bt.reset()
            bt.set_update_mode(k, symmetric_gather=(name == "mirrored"))
            bt.set_strip_flush("always" if name == "strip" else "never")
            bt.set_tuning(0, nt, 0)
            bt.run_known(0, 1 + W)
            st = bt.run_known(1 + W, 1 + W + K, time_kernels=True)
            print(f"k={k:2d} {name:9s} nt={nt}: {st['corrections'] / (st['elapsed_ms'] * 1e-3):10.0f} corr/s, "
                  f"flush {st['rank2_ms'] / st['rank2_launches']:6.2f} ms each x{st['rank2_launches']}, "
                  f"other {st['elapsed_ms'] - st['rank2_ms']:.2f} ms; "
                  f"{16 * (3 + 2 * NL) ** 2 * B / (st['rank2_ms'] / st['rank2_launches']) * 1e-9:.3f} TB/s of Sigma", flush=True)
bt.set_update_mode(0)
bt.close()
