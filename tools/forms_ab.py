"""A/B of the single-filter launch structures ("forms", include/ekfslam.h) on the GPU box, the evidence behind keeping or
retiring a form:
    python tools/forms_ab.py [n ...]                (default n = 60 200 500 700 1000)
For every n: measurement() ticks (known association, V ~ 8 visible landmarks) and data_association() ticks against the
fully surveyed map (J <= 8 readings), each under
    default          call-fused forms (k_call_factors + k_rank2v; k_assoc_call / k_assoc_reading)
    chain            CALL_FUSED off: k_gain + k_rank2 per landmark; k_maha, k_assoc_decide, k_gain, k_rank2 per reading
Round 4's run of this tool (profiles/r04/forms_ab.txt) also timed a third form -- an out-of-place correction launch per
landmark / per reading with a second N x N buffer (ekf_fused.hip, EKF_FORM_FUSED_CORRECTION) -- which lost to `default` at
every size and was deleted on that evidence."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth

NS = [int(x) for x in sys.argv[1:]] or [60, 200, 500, 700, 1000]
FORMS = {"default": capi.FORMS_DEFAULT, "chain": capi.FORMS_DEFAULT & ~capi.FORM_CALL_FUSED}
STEPS, WARM = 260, 60


def tick_known(n, forms):
    ext = 1.5 * np.sqrt(n / 20.0)
    cfg = synth.SimConfig(n=n, steps=STEPS, filters=1, seed=2, half_extent=ext, min_spacing=0.25, max_visible_dis=1.2, vmax=8,
                          v_cmd=0.3, w_cmd=0.2)
    log = synth.make_known_log(cfg)
    inp = [log.expand_step(t) for t in range(STEPS)]
    f = capi.EKF_SLAM(n)
    f.set_forms(forms)
    for t in range(WARM):
        f.prediction(log.twist[t, 0]); f.measurement(*inp[t])
    f.sync()
    t0 = time.perf_counter()
    for t in range(WARM, STEPS):
        f.prediction(log.twist[t, 0]); f.measurement(*inp[t])
    f.sync()
    dt = time.perf_counter() - t0
    corr = int((log.lm_idx[WARM:] >= 0).sum())
    st = f.state
    f.close()
    return dt / (STEPS - WARM) * 1e6, corr / (STEPS - WARM), st


def tick_unknown(n, forms):
    ext = 1.5 * np.sqrt(n / 20.0)
    cfg = synth.SimConfig(n=n, steps=STEPS, filters=1, seed=3, half_extent=ext, min_spacing=0.3, max_visible_dis=1.3, vmax=8,
                          v_cmd=0.3, w_cmd=0.2)
    log = synth.make_unknown_log(cfg)
    meas = [log.meas_xy[t, 0, :log.count[t, 0]] for t in range(STEPS)]
    rng = np.random.default_rng(5)
    rel = synth._robot_frame(log.world, np.zeros((1, 3)))[0]
    f = capi.EKF_SLAM(n)
    f.set_forms(forms)
    f.measurement((rel + rng.normal(0, 0.005, rel.shape)).reshape(-1), np.zeros(n, dtype=np.uint8))
    f.measurement((rel + rng.normal(0, 0.005, rel.shape)).reshape(-1), np.ones(n, dtype=np.uint8))
    kn = np.ones(n, dtype=np.uint8)
    for t in range(WARM):
        f.prediction(log.twist[t, 0]); f.data_association(meas[t], kn)
    f.sync()
    t0 = time.perf_counter()
    nm = 0
    for t in range(WARM, STEPS):
        f.prediction(log.twist[t, 0]); nm += len(f.data_association(meas[t], kn))
    f.sync()
    dt = time.perf_counter() - t0
    st = f.state
    f.close()
    return dt / (STEPS - WARM) * 1e6, nm / (STEPS - WARM), st


for n in NS:
    for what, fn in (("measurement()", tick_known), ("data_association()", tick_unknown)):
        res = {name: fn(n, forms) for name, forms in FORMS.items()}
        same = all(np.array_equal(res["default"][2], r[2]) for r in res.values())
        line = ", ".join(f"{name} {r[0]:7.1f} us" for name, r in res.items())
        print(f"n={n:5d} {what:19s} per tick ({res['default'][1]:.1f} corrections / readings): {line}; bit-identical: {same}", flush=True)
