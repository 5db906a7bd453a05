"""bench.py's delayed leg by itself (default, exact-operand form: paired gain launches, column panel, automatic flush form):
    python tools/delayed_leg.py [B=4096] [k=32] [forms=default]
a warm-up run, then whole flush periods on the same handle (what rocprofv3 / the PMC passes of tools/delayed_pmc.sh wrap)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekf_slam_ml_amd import capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
forms = int(sys.argv[3], 0) if len(sys.argv) > 3 else capi.FORMS_DEFAULT
W, K = 5, 2 * k            # 2 k steps x 2 corrections = 4 whole flush periods
cfg = synth.config5(filters=B, steps=1 + W + K, n=1000)
bt = capi.BatchEKF(B, 1000)
bt.set_forms(forms)
bt.simulate_known_log(cfg, synth.make_world(1000, cfg.half_extent, cfg.min_spacing, cfg.world_seed))
bt.set_update_mode(k)
bt.run_known(0, 1 + W)
st = bt.run_known(1 + W, 1 + W + K, time_kernels=True)
fc = bt.form_counts()
print(f"B={B} k={k} forms={forms:#x}: {st['corrections'] / (st['elapsed_ms'] * 1e-3):.0f} update steps/s, {st['rank2_launches']} flushes "
      f"of {st['rank2_ms'] / max(st['rank2_launches'], 1):.2f} ms, {(st['elapsed_ms'] - st['rank2_ms']) / K * 1e3:.1f} us per step "
      f"outside the flushes; forms {fc}", flush=True)
bt.close()
