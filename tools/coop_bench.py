"""A/B on the GPU box: one-launch cooperative tick (ekf_coop.hip) vs one launch per landmark (ekf_fused.hip) on
BASELINE.json configs[1] (single filter, n = 200, known association) and neighbours, through the C ABI loop."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth


def run(n, log, inputs, steps, coop, wgs, profile=False):
    f = capi.EKF_SLAM(n)
    f.set_call_fused(coop == "cf")
    f.set_cooperative_tick(coop is True, wgs)
    for t in range(50):
        f.prediction(log.twist[t, 0]); f.measurement(*inputs[t])
    f.set_profiling(profile)
    f.sync()
    t0 = time.perf_counter()
    for t in range(50, steps):
        f.prediction(log.twist[t, 0]); f.measurement(*inputs[t])
    f.sync()
    dt = time.perf_counter() - t0
    prof = f.profile() if profile else None
    st = f.state
    f.close()
    return dt, prof, st


def case(n, cfg, wg_list):
    log = synth.make_known_log(cfg)
    steps = cfg.steps
    inputs = [log.expand_step(t) for t in range(steps)]
    corr = int((log.lm_idx[50:] >= 0).sum())
    dt, _, ref = run(n, log, inputs, steps, False, 0)
    _, prof, _ = run(n, log, inputs, steps, False, 0, True)
    print(f"n={n} per-landmark launches : {(steps - 50) / dt:8.0f} steps/s {corr / dt:9.0f} corrections/s  "
          f"{dt / (steps - 50) * 1e6:7.1f} us/step  kernel avg {prof['stream_ms'] / prof['stream_launches'] * 1e3:6.2f} us x {prof['stream_launches']}", flush=True)
    dt, _, st = run(n, log, inputs, steps, "cf", 0)
    _, prof, _ = run(n, log, inputs, steps, "cf", 0, True)
    print(f"n={n} two launches per call  : {(steps - 50) / dt:8.0f} steps/s {corr / dt:9.0f} corrections/s  "
          f"{dt / (steps - 50) * 1e6:7.1f} us/step  stream kernel avg {prof['stream_ms'] / max(prof['stream_launches'], 1) * 1e3:6.2f} us x {prof['stream_launches']}"
          f"  bit-identical {np.array_equal(st, ref)}", flush=True)
    for wgs in wg_list:
        dt, _, st = run(n, log, inputs, steps, True, wgs)
        _, prof, _ = run(n, log, inputs, steps, True, wgs, True)
        print(f"n={n} cooperative tick wgs={wgs:3d}: {(steps - 50) / dt:8.0f} steps/s {corr / dt:9.0f} corrections/s  "
              f"{dt / (steps - 50) * 1e6:7.1f} us/step  kernel avg {prof['stream_ms'] / max(prof['stream_launches'], 1) * 1e3:6.2f} us x {prof['stream_launches']}"
              f"  bit-identical {np.array_equal(st, ref)}", flush=True)


if "ab" in sys.argv: case(200, synth.config2(steps=1000), [0, 16, 25, 32, 50, 100, 200])
if "ab500" in sys.argv:
    c = synth.config2(steps=400); c.n = 500
    case(500, c, [0, 64, 128, 250])


def probe(n=200, wgs=0):
    """fixed cost of a tick (no visible landmark) and marginal cost per hand-off (every landmark visible)"""
    rng = np.random.default_rng(1)
    world = rng.uniform(-4, 4, size=(n, 2))
    sensor = world.reshape(-1).copy()
    for label, vis in (("V=0", np.zeros(n, dtype=np.uint8)), ("V=n", np.ones(n, dtype=np.uint8)),
                       ("V=8", np.r_[np.ones(8, dtype=np.uint8), np.zeros(n - 8, dtype=np.uint8)])):
        for coop in (True, False, "cf"):
            f = capi.EKF_SLAM(n)
            f.set_call_fused(coop == "cf")
            f.set_cooperative_tick(coop is True, wgs)
            f.prediction((0.01, 0.02)); f.measurement(sensor, np.zeros(n, dtype=np.uint8))
            for _ in range(5):
                f.prediction((0.01, 0.02)); f.measurement(sensor, vis)
            f.set_profiling(True)
            reps = 40
            for _ in range(reps):
                f.prediction((0.01, 0.02)); f.measurement(sensor, vis)
            p = f.profile()
            V = int(vis.sum())
            print(f"probe n={n} {label} coop={coop}: {p['stream_ms'] / reps * 1e3:8.1f} us of kernel time per tick"
                  + (f" = {p['stream_ms'] / reps * 1e3 / V:6.2f} us per correction" if V else ""), flush=True)
            f.close()


if "probe" in sys.argv: probe(200)
if "probe" in sys.argv: probe(500)


def trace(n=200, V=6):
    """where the time of one tick goes: per-phase stamps of every workgroup (lane 0, 100 MHz wall clock)"""
    rng = np.random.default_rng(1)
    world = rng.uniform(-4, 4, size=(n, 2))
    sensor = world.reshape(-1).copy()
    vis = np.zeros(n, dtype=np.uint8)
    vis[rng.choice(n, size=V, replace=False)] = 1
    f = capi.EKF_SLAM(n)
    f.set_cooperative_tick(True)
    f.prediction((0.01, 0.02)); f.measurement(sensor, np.zeros(n, dtype=np.uint8))
    f.cooperative_trace(True)
    for _ in range(6):
        f.prediction((0.01, 0.02)); f.measurement(sensor, vis)
    f.sync()
    tr = f.cooperative_trace(True, fetch=True).astype(np.float64)
    f.close()
    cyc = tr[:, 61] - tr[:, 60]
    wall = (tr[:, 63] - tr[:, 0]) / 100.0
    print(f"shader clock during the tick: {np.median(cyc / wall):.0f} cycles per us (s_memtime / s_memrealtime)")
    own = int(np.argmax(tr[:, 50] > 0))
    cy = tr[own, 50:60]
    print("owner of correction 1 (cycles): terms", cy[1] - cy[0], "barrier", cy[2] - cy[1], "G + stores issued", cy[3] - cy[2],
          "drain", cy[4] - cy[3], "barrier", cy[5] - cy[4], "| K", cy[6] - cy[5], "barrier", cy[7] - cy[6], "rows", cy[8] - cy[7], "state+barrier", cy[9] - cy[8])
    oth = (own + 7) % len(tr)
    cy = tr[oth, 50:60]
    print("a consumer of correction 1 (cycles): K", cy[6] - cy[5] if cy[5] else "n/a", "barrier", cy[7] - cy[6], "rows", cy[8] - cy[7], "state+barrier", cy[9] - cy[8])
    t0 = tr[:, 0].min()
    us = (tr - t0) / 100.0
    us[tr == 0] = np.nan
    lms = np.nonzero(vis)[0]
    print(f"trace n={n} V={V} workgroups={len(tr)}: start spread {np.nanmax(us[:, 0]):.2f} us; image loaded (median/max) "
          f"{np.nanmedian(us[:, 1]):.2f}/{np.nanmax(us[:, 1]):.2f}; predicted {np.nanmedian(us[:, 2]):.2f}; readings {np.nanmedian(us[:, 3]):.2f}")
    for v in range(min(V, 9)):
        b = 4 + 6 * v
        ready = us[:, b + 1]
        own = int(np.nanargmin(us[:, b + 2] - 0 * ready)) if False else None
        print(f"  correction {v} (lm {lms[v]}): begin med {np.nanmedian(us[:, b]):6.2f} | flag/terms med {np.nanmedian(us[:, b + 1]):6.2f} min {np.nanmin(us[:, b + 1]):6.2f} "
              f"| block med {np.nanmedian(us[:, b + 2]):6.2f} min {np.nanmin(us[:, b + 2]):6.2f} | K med {np.nanmedian(us[:, b + 3]):6.2f} | rows med {np.nanmedian(us[:, b + 4]):6.2f} max {np.nanmax(us[:, b + 4]):6.2f}")
    print(f"  loop done med {np.nanmedian(us[:, 62]):.2f} max {np.nanmax(us[:, 62]):.2f}; written back med {np.nanmedian(us[:, 63]):.2f} max {np.nanmax(us[:, 63]):.2f}")


if "trace" in sys.argv: trace(200, 6)
if "ab1000" in sys.argv:
    c = synth.config3(steps=300)
    case(1000, c, [])


if "cfonly" in sys.argv:   # for rocprofv3: the default (two launches per call) path alone, n = 200 and n = 1000
    for n, cfg in ((200, synth.config2(steps=600)), (1000, synth.config3(steps=200))):
        log = synth.make_known_log(cfg)
        inputs = [log.expand_step(t) for t in range(cfg.steps)]
        dt, _, _ = run(n, log, inputs, cfg.steps, "cf", 0)
        print(f"n={n}: {dt / (cfg.steps - 50) * 1e6:.1f} us per tick", flush=True)


def cf_trace(n=200, V=6):
    """phase stamps (shader clock) of k_call_factors, workgroup 0: row 0 = control wave (core filter), row 1 = slices"""
    rng = np.random.default_rng(1)
    world = rng.uniform(-4, 4, size=(n, 2))
    sensor = world.reshape(-1).copy()
    vis = np.zeros(n, dtype=np.uint8)
    vis[rng.choice(n, size=V, replace=False)] = 1
    f = capi.EKF_SLAM(n)
    f.prediction((0.01, 0.02)); f.measurement(sensor, np.zeros(n, dtype=np.uint8))
    f.cooperative_trace(True)
    for _ in range(6):
        f.prediction((0.01, 0.02)); f.measurement(sensor, vis)
    f.sync()
    tr = f.cooperative_trace(True, fetch=True).astype(np.float64)
    f.close()
    c, s = tr[0], tr[1]
    t0 = c[0]
    print(f"k_call_factors n={n} V={V} (cycles from kernel start; 2400 cycles = 1 us)")
    print(f"  setup: control {c[1] - t0:.0f}, slices {s[1] - t0:.0f}; terms + gains of correction 0 done {c[2] - t0:.0f}")
    # per correction t (ekf_callfused.hip): barrier | wave 0: terms_h(t+1) || wave 1: core block update(t) || slices: factors
    # + panels(t) | barrier | wave 0: terms_s(t+1) + core gains(t+1)
    for t in range(V):
        b = 3 + 5 * t
        last = t + 1 == V
        print(f"  correction {t}: released {c[b] - t0:8.0f} | control: next terms (h) {c[b + 1] - t0:8.0f} | slices: factors + panels "
              f"{s[b + 1] - t0:8.0f}" + ("" if last else f" | second barrier {c[b + 2] - t0:8.0f} | next S, gains {c[b + 3] - t0:8.0f}"))
    print(f"  end {c[60] - t0:.0f}")


if "cftrace" in sys.argv: cf_trace(200, 6)
