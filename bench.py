#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: EKF update steps/s + achieved HBM GB/s vs roofline, n = 1000.

Workload (config.workload): one GPU's share of BASELINE.json configs[4] -- B independent rigid2d::EKF_SLAM
filters (Monte-Carlo batch), n = 1000 landmarks (N = 2003, fp64), known association, exactly V = 2 landmark
corrections per filter step.  A bench "step" = for every filter: prediction(twist) + measurement(2 readings).
An "EKF update step" (the unit of `value`) = one landmark correction (gain + state + covariance update,
ekf_slam.cpp:137-192); filter steps/s are reported next to it.  Inputs (twists, readings) are uploaded to HBM
before the timed region.  Multi-GPU: one process per GPU, filters sharded by global id, no data-path
collective; RCCL carries only the final throughput reduction ("scaling": "weak").

The same JSON line carries, as separately reported sub-objects (never mixed into `value` / `roofline`):
  delayed_update, active_set_update, unknown_association (+ its full-map variant), small_map_monte_carlo,
  and at N = 1 the other BASELINE.json configurations, each with its own roofline and CPU baseline:
  configs_1 (single filter n = 200 known, through the C ABI), configs_2 (single filter n = 1000 unknown),
  configs_3 (dense F Sigma F^T + Q, N = 10003, fp32 MFMA).

At N > 1 only the contract leg and the delayed leg run (an 8-rank job is ~20 s); the line then also carries the
per-rank spread of the timed region (rank_ms_per_step_min / _max) next to the MAX-reduced ms_per_step.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--filters B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
(`python bench.py --gpus N` with N > 1 outside torchrun starts that launcher as a child process itself.)
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
MFMA_F32_PEAK_TF = 157.3   # same guide: FP32 matrix peak (v_mfma_f32_32x32x2_f32), dense
RANK2_SOURCE = os.path.join(ROOT, "ekf_slam_ml_amd", "csrc", "ekf_kernels.hip")
FULL_MAP_SEED = 36         # configs_2's full-map phase: noise of the two survey calls.  Round 3's seed 33 left one of the 2.4 M
                           # scored pairs 8.9e-7 (relative) from the 10.0 gate; tools/configs2_margins.py: 36 -> 7.2e-5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--filters", type=int, default=0, help="filters per GPU (0 = 4096, reduced to fit HBM)")
    ap.add_argument("--total-filters", type=int, default=0,
                    help="filters of the whole job, sharded over the ranks by global id (blocks may differ by one "
                         "filter); 0 = --filters per GPU (weak scaling)")
    ap.add_argument("--landmarks", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-filters", type=int, default=0)
    ap.add_argument("--delayed-k", type=int, default=32,
                    help="also time the delayed rank-2k update with this many corrections per flush (0 = skip)")
    ap.add_argument("--no-active-set", action="store_true", help="skip the active-set leg")
    ap.add_argument("--no-call-fused", action="store_true", help="skip the call-fused (one pass per measurement() call) leg")
    ap.add_argument("--no-unknown", action="store_true", help="skip the batched unknown-association legs")
    ap.add_argument("--no-small", action="store_true", help="skip the small-map (n = 20) Monte-Carlo leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs[1..3] legs (single filters, dense)")
    ap.add_argument("--only-main", action="store_true", help="the contract leg only (profiling runs)")
    ap.add_argument("--host-log", action="store_true",
                    help="generate the synthetic log on the host (numpy) and upload it, instead of on the device")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--nt", type=int, default=-1)
    a = ap.parse_args()
    if a.only_main:
        a.no_active_set = a.no_unknown = a.no_small = a.no_configs = a.no_call_fused = True
        a.delayed_k = 0
    return a


LINE_BUDGET = 7500   # characters of the one JSON line (tests/test_host.py holds the committed line to it, too)
_FULL_PRECISION = {"value", "ms_per_step", "achieved", "frac", "avg_launch_ms", "rank_ms_per_step_min", "rank_ms_per_step_max",
                   "algorithmic_bytes_per_launch", "traffic", "declared_bytes_per_correction"}


# computed (and used by the checks inside bench.py) but not printed: derivable from what is
_DROP = {"scores_per_s", "measurements", "corrections", "score_launches", "steps_per_flush", "covariance_GBps", "parity_filters",
         "known_landmarks_min", "corrections_per_launch", "kernel_launches", "fp64_check_rows", "touched_landmarks_max",
         "hbm_bytes_per_gpu", "state_dim", "peak_nested", "achieved_GBps_on_declared_bytes", "launches_nested",
         "active_dimension_max", "measurement_slots", "steps_timed", "corrections_per_step"}


def compact(o, key=None, depth=0):
    """Numbers only, short: every float below the top-level contract keys is rounded to 5 significant digits, the
    Monte-Carlo blocks become [nees_mean, nees_max, rmse_xy, rmse_theta, mean_trace_pose_cov, frac_below_95pct(, step)]
    lists.  The prose that used to travel in `note` / `sample` / `workload` lives in README.md ("Reading the bench line")."""
    if isinstance(o, dict):
        if "nees_mean" in o:
            return [compact(v, None, depth + 1) for v in o.values()]
        return {k: compact(v, k, depth + 1) for k, v in o.items() if not (depth >= 1 and k in _DROP)}
    if isinstance(o, (list, tuple)):
        return [compact(v, None, depth + 1) for v in o]
    if isinstance(o, float):
        if o != o or o in (float("inf"), float("-inf")):
            return None
        if depth <= 2 and key in _FULL_PRECISION:
            return float(f"{o:.9g}")
        return float(f"{o:.5g}")
    return o


def sha256_of(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def pmc_traffic(kernel, B, n):
    """HBM bytes per launch of the dominant kernel from the committed PMC record (profiles/rank2_traffic.json), or
    None when that record was taken on a different kernel instantiation, another build of its source file, or another
    pool shape -- a stale figure is never quoted."""
    tfile = os.path.join(ROOT, "profiles", "rank2_traffic.json")
    try:
        tj = json.load(open(tfile))
    except (OSError, ValueError):
        return None, "no profiles/rank2_traffic.json"
    name = "".join(str(tj.get("kernel", "")).split())
    if "".join(kernel.split()) not in name:
        return None, f"PMC record is for {tj.get('kernel')!r}, this run launches {kernel}"
    if tj.get("source_sha256") != sha256_of(RANK2_SOURCE):
        return None, "PMC record was taken on another build of ekf_kernels.hip"
    if tj.get("filters") != B or tj.get("n") != n:
        return None, "PMC record was taken on another pool shape"
    return tj.get("hbm_bytes_per_launch"), None


def cpu_baseline(sub, K, t_warm, cores, gpu_state):
    """The CPU checker's structured restatement ("port"), OpenMP over filters, on a bounded sample of the
    same log (its first filters), timed on this box's host cores.  Doubles as a parity spot-check."""
    import numpy as np
    from oracle import binding as ob  # checker / baseline only -- never on the product path
    st, _, stats = ob.batch_run_known(sub, ob.STRUCTURED, t_warm=t_warm, nthreads=cores, want_cov=False, fast=True)
    B = sub.twist.shape[1]
    return {"value": stats["corrections"] / stats["seconds"], "unit": "update steps/s", "cores": stats["threads"],
            "kind": "port",
            "sample": f"{B} filters (strided over the pool) x {K} steps x 2 corr, n={sub.cfg.n}, {stats['seconds']:.1f} s, "
                      f"structured C port, OpenMP",
            "max_abs_state_diff_vs_gpu": float(np.abs(st - gpu_state).max())}


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[1]: one filter, n = 200, known association, through the C ABI (the node's call sequence:
# prediction(twist) + measurement(sensor_reading, visible_list) per 10 Hz tick; slam.cpp:433-434).
# ------------------------------------------------------------------------------------------------------------------
def leg_configs_1(device, cores, steps=2000, warm=100, want_cpu=True):
    import numpy as np
    from ekf_slam_ml_amd import capi, synth
    cfg = synth.config2(steps=steps)
    log = synth.make_known_log(cfg)
    n, N = cfg.n, 3 + 2 * cfg.n
    inputs = [log.expand_step(t) for t in range(steps)]
    bytes_corr = 16.0 * N * N

    def run(profile):
        f = capi.EKF_SLAM(n, device=device)
        for t in range(warm):
            f.prediction(log.twist[t, 0]); f.measurement(*inputs[t])
        f.set_profiling(profile)
        f.sync()
        t0 = time.perf_counter()
        for t in range(warm, steps):
            f.prediction(log.twist[t, 0]); f.measurement(*inputs[t])
        f.sync()
        dt = time.perf_counter() - t0
        prof = f.profile() if profile else None
        st = f.state
        f.close()
        return dt, prof, st

    run(False)   # untimed pass: first launches of every kernel instantiation of the path (code-object loading, clocks)
    dt, _, st_gpu = run(False)
    _, prof, _ = run(True)
    corr = int((log.lm_idx[warm:] >= 0).sum())
    out = {"workload": f"configs[1]: 1 filter n={n} known, {steps - warm} steps via the C ABI",
           "value": corr / dt, "unit": "update steps/s", "filter_steps_per_s": (steps - warm) / dt,
           "corrections_per_step": corr / (steps - warm), "us_per_correction": dt / corr * 1e6}
    kern_us = prof["stream_ms"] / max(prof["stream_launches"], 1) * 1e3
    # Two figures, never mixed: `contract` prices every correction at the eager stream's
    # 16 N^2 bytes; `streamed` counts the bytes the launches of this path really move -- one pass over Sigma per CALL
    # (16 N^2 per k_rank2v launch) plus the factor kernel's two panels (2 x (3 + 2V) rows / columns of 8 N bytes) and the
    # factor rows it writes and the pass re-reads (4 rows x 8 N bytes x 2 per correction).
    launches = max(prof["stream_launches"], 1)
    streamed_bytes = launches * 16.0 * N * N + (steps - warm) * 2 * 3 * 8.0 * N + corr * (2 * 2 * 8.0 * N + 2 * 4 * 8.0 * N)
    out["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                       "achieved": streamed_bytes / dt / 1e9, "frac": streamed_bytes / dt / 1e9 / HBM_PEAK_GBS,
                       "contract_frac": corr / dt * bytes_corr / 1e9 / HBM_PEAK_GBS,
                       "traffic": None, "kernel_avg_us": kern_us, "kernel_launches": prof["stream_launches"],
                       "corrections_per_launch": corr / launches,
                       "in_kernel_GBps": bytes_corr / (kern_us * 1e-6) / 1e9}
    if want_cpu:
        from oracle import binding as ob  # checker / baseline only
        o = ob.OracleEKF(n, ob.STRUCTURED, fast=True)
        t0 = time.perf_counter()
        for t in range(steps):
            o.prediction(*log.twist[t, 0]); o.measurement(*inputs[t])
        cdt = time.perf_counter() - t0
        call = int((log.lm_idx >= 0).sum())
        od = ob.OracleEKF(n, ob.DENSE, fast=True)
        dsteps = 6
        t0 = time.perf_counter()
        for t in range(dsteps):
            od.prediction(*log.twist[t, 0]); od.measurement(*inputs[t])
        ddt = time.perf_counter() - t0
        dcorr = int((log.lm_idx[:dsteps] >= 0).sum())
        out["cpu_baseline"] = {"value": call / cdt, "unit": "update steps/s", "cores": 1, "kind": "port",
                               "sample": f"same {steps} steps, {cdt:.2f} s, structured C port",
                               "dense_literal": {"value": dcorr / max(ddt, 1e-9), "sample": f"{dsteps} steps, {ddt:.1f} s, O(N^3) loops"},
                               "max_abs_state_diff_vs_gpu": float(np.abs(o.state - st_gpu).max())}
    return out


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2]: one filter, n = 1000, unknown association (prediction + data_association per tick;
# unknown_data_assoc.cpp:414-415).  (a) the discovery run as SURVEY 8(d) specifies it (M grows from 0);
# (b) the same sensing against a FULLY discovered map (M = 1000 from the first measurement): every measurement is
# scored against all 1000 landmarks (ekf_slam.cpp:300-309) and the winner corrected at full width (:331-390).
# ------------------------------------------------------------------------------------------------------------------
def leg_configs_2(device, cores, steps=2000, full_steps=300, want_cpu=True):
    import numpy as np
    from ekf_slam_ml_amd import capi, synth
    cfg = synth.config3(steps=steps)
    log = synth.make_unknown_log(cfg)
    n, N = cfg.n, 3 + 2 * cfg.n
    meas = [log.meas_xy[t, 0, :log.count[t, 0]] for t in range(steps)]
    out = {"workload": f"configs[2]: 1 filter n={n} unknown, J<=8 per step via the C ABI"}

    # (a) discovery run (an untimed pass first: the path changes kernels as the map grows -- LDS-resident step kernel,
    # then two launches per reading -- and the first launch of every instantiation loads its code object)
    warm = 50

    def discover():
        f = capi.EKF_SLAM(n, device=device)
        known = np.zeros(n, dtype=np.uint8)
        for t in range(warm):
            f.prediction(log.twist[t, 0]); f.data_association(meas[t], known)
        f.sync()
        t0 = time.perf_counter()
        nm = nc = 0
        scores = 0
        decs = []
        for t in range(warm, steps):
            kc = int(known.sum())
            f.prediction(log.twist[t, 0]); a = f.data_association(meas[t], known)
            nm += len(a); nc += int((a >= 0).sum())
            scores += len(a) * kc  # lower bound (known_count grows inside the call)
            decs.append(a)
        f.sync()
        dt = time.perf_counter() - t0
        st = f.state
        f.close()
        return dt, nm, nc, scores, known, decs, st

    discover()
    dt, nm, nc, scores, known, gpu_decs, gpu_disc_state = discover()
    out["discovery"] = {"value": (steps - warm) / dt, "unit": "filter steps/s", "measurements_per_s": nm / dt,
                        "corrections_per_s": nc / dt, "scores_per_s": scores / dt, "known_landmarks_end": int(known.sum()),
                        "steps": steps - warm}

    # (b) full map: phase A builds the map through the known-association API (first call initialises all n landmarks,
    # second call corrects all n at full width), phase B is data_association with known_list all true.
    rng = np.random.default_rng(FULL_MAP_SEED)
    world = log.world
    pose0 = np.zeros(3)

    def all_readings():
        rel = synth._robot_frame(world, pose0.reshape(1, 3))[0]
        return (rel + rng.normal(0.0, cfg.sensor_std, size=rel.shape)).reshape(-1)

    def build(profile):
        g = capi.EKF_SLAM(n, device=device)
        g.measurement(all_readings(), np.zeros(n, dtype=np.uint8))
        g.measurement(all_readings(), np.ones(n, dtype=np.uint8))
        g.set_profiling(profile)
        g.sync()
        return g

    rng = np.random.default_rng(FULL_MAP_SEED); g = build(False)
    for t in range(10):   # untimed, on a throw-away object: first launches of the full-map kernels
        g.prediction(log.twist[t, 0]); g.data_association(meas[t], np.ones(n, dtype=np.uint8))
    g.close()
    rng = np.random.default_rng(FULL_MAP_SEED); g = build(False)
    snap_state, snap_cov = (g.state, g.cov) if want_cpu else (None, None)
    kn = np.ones(n, dtype=np.uint8)
    t0 = time.perf_counter()
    nm = nc = 0
    full_decs = []
    for t in range(full_steps):
        g.prediction(log.twist[t, 0]); a = g.data_association(meas[t], kn)
        nm += len(a); nc += int((a >= 0).sum())
        full_decs.append(a)
    g.sync()
    dt = time.perf_counter() - t0
    st_gpu = g.state
    g.close()
    rng = np.random.default_rng(FULL_MAP_SEED); g = build(True)
    for t in range(full_steps):
        g.prediction(log.twist[t, 0]); g.data_association(meas[t], kn)
    prof = g.profile()
    g.close()
    bytes_corr = 16.0 * N * N
    k_us = prof["stream_ms"] / max(prof["stream_launches"], 1) * 1e3
    s_us = prof["score_ms"] / max(prof["score_launches"], 1) * 1e3
    out["full_map"] = {"value": nm / dt, "unit": "measurements/s",
                       "filter_steps_per_s": full_steps / dt, "scores_per_s": nm * n / dt, "corrections_per_s": nc / dt,
                       "known_landmarks": n, "measurements": nm, "corrections": nc, "steps": full_steps,
                       "score_kernel_avg_us": s_us, "score_launches": prof["score_launches"],
                       "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                    "achieved": prof["stream_launches"] * bytes_corr / dt / 1e9,
                                    "frac": prof["stream_launches"] * bytes_corr / dt / 1e9 / HBM_PEAK_GBS,
                                    "contract_frac": nc / dt * bytes_corr / 1e9 / HBM_PEAK_GBS,
                                    "traffic": None, "kernel_avg_us": k_us, "kernel_launches": prof["stream_launches"],
                                    "in_kernel_GBps": bytes_corr / (k_us * 1e-6) / 1e9}}
    if want_cpu:
        from oracle import binding as ob  # checker / baseline only
        o = ob.OracleEKF(n, ob.STRUCTURED, fast=True)
        o.state, o.cov = snap_state, snap_cov
        o.set_init_flag(1)
        ko = np.ones(n, dtype=np.uint8)
        csteps = full_steps
        # decision margins of the run (the checker records, for every scored (reading, landmark) pair, the relative distance
        # of the score to the gates 10.0 / 1.0 of ekf_slam.cpp:293,330 and the winner-to-runner-up gap, :305-309): the only
        # place where another summation order could change a result by more than rounding
        mg = ob.new_margins()
        t0 = time.perf_counter()
        cm = 0
        same = True
        for t in range(csteps):
            o.prediction(*log.twist[t, 0]); b = o.data_association(meas[t], ko, mg)
            cm += len(b)
            same = same and bool(np.array_equal(b, full_decs[t]))
        cdt = time.perf_counter() - t0
        out["full_map"]["min_gate_margin"] = float(mg[:3].min())
        out["full_map"]["min_decision_margin"] = float(mg[4])
        out["full_map"]["decisions_identical_to_cpu_port"] = same
        # ... and of the discovery run (replayed on the checker from the start)
        od = ob.OracleEKF(n, ob.STRUCTURED, fast=True)
        kd = np.zeros(n, dtype=np.uint8)
        mgd = ob.new_margins()
        same = True
        for t in range(steps):
            od.prediction(*log.twist[t, 0]); b = od.data_association(meas[t], kd, mgd)
            if t >= warm:
                same = same and bool(np.array_equal(b, gpu_decs[t - warm]))
        out["discovery"]["min_gate_margin"] = float(mgd[:3].min())
        out["discovery"]["min_decision_margin"] = float(mgd[4])
        out["discovery"]["decisions_identical_to_cpu_port"] = same
        out["discovery"]["max_abs_state_diff_vs_cpu_port"] = float(np.abs(od.state - gpu_disc_state).max())
        out["full_map"]["cpu_baseline"] = {
            "value": cm / cdt, "unit": "measurements/s", "cores": 1, "kind": "port",
            "sample": f"same {csteps} steps, {cdt:.1f} s, structured C port"}
        if csteps == full_steps:
            out["full_map"]["cpu_baseline"]["max_abs_state_diff_vs_gpu"] = float(np.abs(o.state - st_gpu).max())
    return out


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3]: dense general-F propagation Sigma <- F Sigma F^T + Q, n = 5000 (N = 10003), fp32 MFMA.
# ------------------------------------------------------------------------------------------------------------------
def leg_configs_3(device, cores, N=10003, iters=9, want_cpu=True):
    import numpy as np
    from ekf_slam_ml_amd import capi
    rng = np.random.default_rng(4)
    F = np.eye(N, dtype=np.float32) + rng.standard_normal((N, N), dtype=np.float32) * np.float32(0.05 / np.sqrt(N))
    A = rng.standard_normal((N, 64), dtype=np.float32)
    S = A @ A.T / np.float32(64) + np.eye(N, dtype=np.float32)
    Q = np.zeros((N, N), dtype=np.float32)
    Q[0, 0] = Q[1, 1] = Q[2, 2] = 1e-4
    d = capi.DensePropagator(N, device=device)
    info = d.launch_info()
    d.set(F, S, Q)
    d.propagate(1)  # warm-up + the result that is checked
    got = d.sigma
    d.propagate(3)  # untimed: the first products after an idle phase run below the sustained clock (16-18 ms against 15)
    rows = np.array(sorted({0, 2, N // 2, N - 1, *[int(x) for x in rng.integers(0, N, size=4)]}))
    F64 = F.astype(np.float64)
    want = (F64[rows] @ S.astype(np.float64)) @ F64.T + Q[rows].astype(np.float64)
    err = float(np.abs(got[rows] - want).max() / np.abs(want).max())
    d.set(F, S, Q)
    ms = [d.propagate(1) for _ in range(iters)]
    d.close()
    med = float(np.median(ms))
    flop = 4.0 * float(N) ** 3
    tf = flop / (med * 1e-3) / 1e12
    out = {"workload": f"configs[3]: dense F Sigma F^T + Q, N={N}, fp32 MFMA",
           "value": 1e3 / med, "unit": "propagations/s", "ms_per_propagation": med, "dtype": "f32",
           "launch": info, "fp64_check_rel_err": err, "fp64_check_rows": len(rows),
           "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_F32_PEAK_TF, "achieved": tf,
                        "frac": tf / MFMA_F32_PEAK_TF, "traffic": None, "avg_launch_ms": med / 2.0}}
    if want_cpu:
        try:
            from threadpoolctl import threadpool_limits
        except ImportError:
            threadpool_limits = None
        Nc = 3072
        Fc, Sc = np.ascontiguousarray(F[:Nc, :Nc]), np.ascontiguousarray(S[:Nc, :Nc])

        def prod():
            return (Fc @ Sc) @ Fc.T + Q[:Nc, :Nc]
        import contextlib
        with (threadpool_limits(limits=cores) if threadpool_limits else contextlib.nullcontext()):
            prod()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                prod()
            cdt = (time.perf_counter() - t0) / reps
        ctf = 4.0 * Nc ** 3 / cdt / 1e12
        out["cpu_baseline"] = {"value": ctf * 1e12 / flop, "unit": "propagations/s (scaled by flop)",
                               "cores": cores, "kind": "port", "tflops": ctf,
                               "sample": f"N={Nc}, {cdt:.2f} s, two OpenBLAS sgemm"}
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the one-process-per-GPU launcher as a CHILD (nothing has touched the GPU
        # yet in this process; never an exec after GPU initialisation) and exit with its code
        port = str(29500 + os.getpid() % 1000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); pass --gpus {world}")
    dist = None
    import numpy as np
    # torch first: it bundles its own libamdhip64.so.7 and libekfslam_hip.so must share that runtime
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path)")
    # one process per GPU; EKF_DIST_BACKEND=gloo is a rehearsal mode that lets several ranks share one GPU
    # (RCCL refuses two ranks on one device) -- the reduction then runs on CPU tensors
    backend = os.environ.get("EKF_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    red_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local)
    # Under the one-process-per-GPU launcher (RANK / MASTER_* in the environment) the process group is created even for a
    # single rank: `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` then takes the very path an 8-rank job
    # takes -- init_process_group("nccl", device_id=...), barriers and the scalar all-reduces through RCCL.
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from ekf_slam_ml_amd import capi, synth

    n = a.landmarks
    N = 3 + 2 * n
    ld = capi.leading_dimension(n)
    per_filter = N * ld * 8 + 6 * ld * 8 + 4 * n * 8
    free, total = torch.cuda.mem_get_info(local)
    B = a.filters if a.filters > 0 else 4096
    B = max(1, min(B, int(0.90 * free // per_filter), 65535))
    K, W = a.steps, a.warmup
    # The delayed leg times whole flushes only: its step count Kd is K rounded up until the 2*Kd corrections per filter
    # are a multiple of the k corrections per flush (a trailing near-empty flush would dilute the figure).
    Kd = K
    if a.delayed_k > 0:
        while (2 * Kd) % a.delayed_k:
            Kd += 1
    T = W + max(K, Kd) + 1  # step 0 = first measurement() call (landmark initialisation, no corrections)

    from ekf_slam_ml_amd import shard
    # weak scaling: B filters per GPU, contiguous blocks of global ids; --total-filters shards a fixed job instead
    # (blocks then differ by at most one filter)
    first_id, count = shard.shard(a.total_filters if a.total_filters > 0 else B * world, world, rank)
    if count > B or count < 1:
        raise SystemExit(f"bench.py: rank {rank}'s share of {count} filters does not fit ({B} per GPU)")
    B = count
    cfg = synth.config5(filters=B, steps=T, first_filter_id=first_id, n=n)
    bt = capi.BatchEKF(B, n, device=local)
    bt.set_tuning(a.rows, a.nt)
    # The contract leg is the EAGER per-landmark stream: every correction streams Sigma once (16 N^2 B, SURVEY.md
    # section 8(d)).  The library's default for pools is the exact call-fused pass (one stream per call), which is
    # reported separately below -- so the default is switched off here, explicitly.
    bt.set_call_fused(False)
    if a.host_log:
        log = synth.make_known_log(cfg)
        bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
    else:
        # inputs are generated ON THE DEVICE (same noise model and random-number addressing as synth.py)
        world_xy = synth.make_world(n, cfg.half_extent, cfg.min_spacing, cfg.world_seed)
        bt.simulate_known_log(cfg, world_xy)
        log = None
    bt.run_known(0, 1 + W)  # init step + W untimed warm-up steps

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    st = bt.run_known(1 + W, 1 + W + K, time_kernels=True)  # returns after the stream has drained
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    fence()
    wall = t1 - t0

    corr = float(st["corrections"])
    fsteps = float(st["filter_steps"])
    # the only collectives of the job: RCCL all-reduces of a few scalars (max / min wall, summed work, ranks seen)
    wall_min, wall_max = shard.wall_spread(wall, device=red_dev)
    wall, corr, fsteps = shard.reduce_throughput(wall, corr, fsteps, device=red_dev)
    ranks_seen = shard.count_ranks(device=red_dev)
    # Monte-Carlo consistency of the batch against the simulated ground truth (f4), at the step the pool stands at
    # NOW: the last step it ran is W + K (step 0 = the initialising call)
    mc_main = bt.mc_stats(W + K) if (rank == 0 and not a.host_log) else None
    if world > 1:   # an N-rank job runs the contract leg and the delayed leg only
        a.no_active_set = a.no_unknown = a.no_small = a.no_configs = a.no_call_fused = True
    cores = int(os.environ.get("EKF_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
    want_cpu = world == 1 and not a.no_cpu_baseline
    # the eager leg's own end states: the CPU baseline's parity spot-check and the other legs compare against THESE
    # (a STRIDED sample over the whole pool -- first and last filter included -- so that no slab of the 132-GB covariance
    # pool goes unchecked; round 3 took the first Bc filters)
    Bc = a.cpu_filters if a.cpu_filters > 0 else min(B, 160 * cores)
    cpu_ids = np.unique(np.linspace(0, B - 1, Bc).round().astype(np.int64))
    Bc = len(cpu_ids)
    eager_state = eager_first = None
    if rank == 0:
        eager_first = np.stack([bt.state(b) for b in range(min(B, 4))])   # what the other legs compare against
        if want_cpu:
            eager_state = np.stack([bt.state(int(b)) for b in cpu_ids])

    # Second, separately reported leg: the same log with the delayed
    # rank-2k covariance update.  Its traffic is different by construction, so it has its own declared
    # bytes and never enters `value` / `roofline` above.
    delayed = None
    if a.delayed_k > 0:
        bt.reset()
        bt.set_update_mode(a.delayed_k)
        bt.run_known(0, 1 + W)
        fence()
        t0 = time.perf_counter()
        sd = bt.run_known(1 + W, 1 + W + Kd, time_kernels=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fence()
        dwall, dcorr, _ = shard.reduce_throughput(t1 - t0, float(sd["corrections"]), float(sd["filter_steps"]),
                                                  device=red_dev)
        if rank == 0:
            kk = float(a.delayed_k)
            Nf = float(N)
            # declared algorithmic bytes per correction: flush share (Sigma read+write + factor reads) + average
            # factor read of the gain step + base gathers + factor/state append.  The two corrections of a step share
            # ONE launch (k_gain_delayed_pair): the pending factors (0, 4, ..., 2k - 4 vectors of 16 N bytes: k - 2 on
            # average) and the 7 rows + 7 columns of the two landmarks are read once per PAIR; 4 rows of 8 N bytes are
            # appended per correction.  (One launch per landmark: 16 N (k - 1) + 112 N for these terms.)
            per_corr = (16.0 * Nf * Nf + 32.0 * Nf * kk) / kk + 0.5 * 16.0 * Nf * (kk - 2.0) + (0.5 * 14.0 + 4.0) * 8.0 * Nf
            # With the kept current rows / columns (EKF_FORM_CURRENT_COLUMNS, the default) a paired gain launch no longer reads
            # all pending factors: it reads and rewrites the 14 kept vectors of its 7 core indices, reads the 8 factor rows
            # appended since the last launch, appends 8 rows and moves the state: 46 x 8 N bytes per PAIR in the steady state
            # (a landmark met for the first time since the flush starts from the stored entries and all pending factors).
            per_corr_kept = (16.0 * Nf * Nf + 32.0 * Nf * kk) / kk + 0.5 * 46.0 * 8.0 * Nf
            delayed = {"value": dcorr / dwall, "unit": "update steps/s", "corrections_per_flush": a.delayed_k,
                       "steps": Kd, "ms_per_step": dwall / Kd * 1e3, "flushes": sd["rank2_launches"],
                       "flush_avg_ms": sd["rank2_ms"] / max(sd["rank2_launches"], 1),
                       "flush_share_of_time": sd["rank2_ms"] / sd["elapsed_ms"],
                       "declared_bytes_per_correction": per_corr,
                       "achieved_GBps_on_declared_bytes": dcorr / world * per_corr / dwall / 1e9,
                       "frac_of_8TBps": dcorr / world * per_corr / dwall / 1e9 / HBM_PEAK_GBS,
                       "speedup_vs_eager": (dcorr / dwall) / (corr / wall)}
            # Parity of THIS configuration (n, k, pairs, automatic flush form) at the step it stands at, W + Kd: the first
            # filters of the pool (same global ids -> same inputs) re-run eagerly on a side pool, and on the CPU port.
            nref = min(B, 4)
            dstate = np.stack([bt.state(b) for b in range(nref)])
            dcov = bt.cov(0)
            import copy
            cfg_r = copy.copy(cfg)
            cfg_r.filters = nref
            ref = capi.BatchEKF(nref, n, device=local)
            ref.set_call_fused(False)
            if a.host_log:
                ref.upload_known_log(log.twist[:, :nref], log.lm_idx[:, :nref], log.z_xy[:, :nref], log.init_xy[:nref])
            else:
                ref.simulate_known_log(cfg_r, world_xy)
            ref.run_known(0, 1 + W + Kd)
            rstate = np.stack([ref.state(b) for b in range(nref)])
            rcov = ref.cov(0)
            delayed["parity_filters"] = nref
            delayed["parity_step"] = W + Kd
            delayed["max_abs_state_diff_vs_eager"] = float(np.abs(dstate - rstate).max())
            delayed["max_rel_cov_diff_vs_eager"] = float(np.abs(dcov - rcov).max() / np.abs(rcov).max())
            fcn = bt.form_counts()
            delayed["flush_form"] = "k_flush_strip" if fcn["flush_strip"] else "k_flush"
            delayed["gain_launches_from_column_panel"] = fcn["gain_from_panel"]
            if fcn["gain_from_panel"] and (bt.forms & capi.FORM_CURRENT_COLUMNS):
                delayed["declared_bytes_per_correction"] = per_corr_kept
                delayed["frac_of_8TBps"] = dcorr / world * per_corr_kept / dwall / 1e9 / HBM_PEAK_GBS
                delayed["declared_bytes_rebuilt_form"] = per_corr   # (round 3's formula: every launch rebuilds from all pending factors)
            delayed["gain_launches_paired"] = fcn["gain_pairs"]
            if want_cpu:
                from oracle import binding as ob  # checker only
                if a.host_log:
                    subr = synth.KnownLog(cfg_r, log.world, log.twist[:, :nref], log.lm_idx[:, :nref], log.z_xy[:, :nref],
                                          log.init_xy[:nref])
                else:
                    tw, li, zz, ii, _ = ref.download_log(want_truth=False)
                    subr = synth.KnownLog(cfg_r, world_xy, tw, li, zz, ii)
                cfg_r2 = copy.copy(cfg_r)
                cfg_r2.steps = 1 + W + Kd
                subr = synth.KnownLog(cfg_r2, subr.world, subr.twist[:1 + W + Kd], subr.lm_idx[:1 + W + Kd],
                                      subr.z_xy[:1 + W + Kd], subr.init_xy)
                cst, ccv, _ = ob.batch_run_known(subr, ob.STRUCTURED, nthreads=min(cores, nref), want_cov=True, fast=False)
                delayed["max_abs_state_diff_vs_cpu_port"] = float(np.abs(dstate - cst).max())
                delayed["max_rel_cov_diff_vs_cpu_port"] = float(np.abs(dcov - ccv[0]).max() / np.abs(ccv[0]).max())
            ref.close()
        # the same leg with the opt-in symmetric gather (the gain step reads Sigma(c, r) for Sigma(r, c): coalesced rows
        # instead of 16-KB-strided column entries; still within the mode's 1e-9, see ekf_set_update_mode)
        bt.reset()
        bt.set_update_mode(a.delayed_k, symmetric_gather=True)
        bt.run_known(0, 1 + W)
        fence()
        t0 = time.perf_counter()
        sg = bt.run_known(1 + W, 1 + W + Kd, time_kernels=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fence()
        gwall, gcorr, _ = shard.reduce_throughput(t1 - t0, float(sg["corrections"]), float(sg["filter_steps"]), device=red_dev)
        if rank == 0:
            gstate = np.stack([bt.state(b) for b in range(nref)])
            gcov = bt.cov(0)
            delayed["symmetric"] = {"value": gcorr / gwall, "unit": "update steps/s",
                                    "flush_avg_ms": sg["rank2_ms"] / max(sg["rank2_launches"], 1),
                                    "flush_form": "k_flush_sym" if bt.form_counts()["flush_mirrored"] else "full",
                                    "max_abs_state_diff_vs_eager": float(np.abs(gstate - rstate).max()),
                                    "max_rel_cov_diff_vs_eager": float(np.abs(gcov - rcov).max() / np.abs(rcov).max())}
        # ... and the default form at the largest k the strip flush serves (40 corrections = 80 pending vectors): since the gain
        # launch no longer grows with the pending count (kept current rows / columns), fewer, fuller flushes are what is left
        # to gain.  Its own step count (whole flush periods); parity is the k = 32 leg's business.
        k2 = 40
        if a.delayed_k == 32 and T - (1 + W) >= k2 // 2:
            K2 = (T - (1 + W)) // (k2 // 2) * (k2 // 2)
            bt.reset()
            bt.set_update_mode(k2)
            bt.run_known(0, 1 + W)
            fence()
            t0 = time.perf_counter()
            s2 = bt.run_known(1 + W, 1 + W + K2, time_kernels=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fence()
            w2, c2, _ = shard.reduce_throughput(t1 - t0, float(s2["corrections"]), float(s2["filter_steps"]), device=red_dev)
            if rank == 0:
                delayed["k40"] = {"value": c2 / w2, "steps": K2, "flushes": s2["rank2_launches"],
                                  "flush_avg_ms": s2["rank2_ms"] / max(s2["rank2_launches"], 1)}
        bt.set_update_mode(0)

    # Separately reported leg: the SAME steps with every measurement() call fused (ekf_callfused.hip) -- the gains and
    # H*Sigma rows of both corrections of a call come from thin panels of Sigma, then ONE pass over Sigma applies them in
    # order: bit-identical to the eager leg, 16 N^2 bytes per CALL instead of per correction.  Declared bytes per
    # correction: 16 N^2 / V + the panel reads and factor traffic of the factor kernel.
    callf = None
    if not a.no_call_fused:
        bt.reset()
        bt.set_call_fused(True)
        bt.run_known(0, 1 + W)
        fence()
        t0 = time.perf_counter()
        sc = bt.run_known(1 + W, 1 + W + K, time_kernels=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fence()
        cwall, ccorr, _ = shard.reduce_throughput(t1 - t0, float(sc["corrections"]), float(sc["filter_steps"]),
                                                  device=red_dev)
        if rank == 0:
            cstate = np.stack([bt.state(b) for b in range(min(B, 4))])
            Vc = 2.0
            Nf = float(N)
            per_corr = 16.0 * Nf * Nf / Vc + (2.0 * (3.0 + 2.0 * Vc) * 8.0 * Nf + 4.0 * 8.0 * Nf * 2.0) / Vc
            pass_s = sc["rank2_ms"] / max(sc["rank2_launches"], 1) * 1e-3
            callf = {"value": ccorr / cwall, "unit": "update steps/s", "ms_per_step": cwall / K * 1e3,
                     "passes": sc["rank2_launches"], "pass_avg_ms": pass_s * 1e3,
                     "pass_GBps": sc["rank2_bytes_per_launch"] / pass_s / 1e9,
                     "pass_frac_of_8TBps": sc["rank2_bytes_per_launch"] / pass_s / 1e9 / HBM_PEAK_GBS,
                     "pass_share_of_time": sc["rank2_ms"] / sc["elapsed_ms"],
                     "declared_bytes_per_correction": per_corr,
                     "achieved_GBps_on_declared_bytes": ccorr / world * per_corr / cwall / 1e9,
                     "frac_of_8TBps": ccorr / world * per_corr / cwall / 1e9 / HBM_PEAK_GBS,
                     "speedup_vs_eager": (ccorr / cwall) / (corr / wall),
                     "bit_identical_to_eager": bool(np.array_equal(cstate, eager_first[:len(cstate)]))}
        bt.set_call_fused(False)

    # Third, separately reported leg: the eager correction restricted to the rows of the TOUCHED set (exact,
    # bit-identical; SURVEY.md section 7 "exact active-set sparsity").  Its cost depends on how many landmarks
    # a filter has corrected so far (here <= 2 per step), not on n -- it says nothing about the dense roofline.
    active = None
    if not a.no_active_set:
        bt.reset()
        bt.set_active_set(True)
        bt.run_known(0, 1 + W)
        fence()
        t0 = time.perf_counter()
        sa = bt.run_known(1 + W, 1 + W + K, time_kernels=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fence()
        awall, acorr, _ = shard.reduce_throughput(t1 - t0, float(sa["corrections"]), float(sa["filter_steps"]),
                                                  device=red_dev)
        if rank == 0:
            astate = np.stack([bt.state(b) for b in range(min(B, 4))])
            tch = bt.touched()
            active = {"value": acorr / awall, "unit": "update steps/s", "ms_per_step": awall / K * 1e3,
                      "touched_landmarks_mean": float(tch.mean()), "touched_landmarks_max": int(tch.max()),
                      "declared_bytes_per_correction": 16.0 * N * (3 + 2 * float(tch.mean())),
                      "rank2_share_of_time": sa["rank2_ms"] / sa["elapsed_ms"],
                      "speedup_vs_eager": (acorr / awall) / (corr / wall),
                      "bit_identical_to_eager": bool(np.array_equal(astate, eager_first[:len(astate)]))}
        bt.set_active_set(False)

    out = None
    if rank == 0:
        r2_avg_s = st["rank2_ms"] / max(st["rank2_launches"], 1) * 1e-3
        achieved = st["rank2_bytes_per_launch"] / r2_avg_s / 1e9
        kname, krows = bt.rank2_kernel()
        traffic, why_not = pmc_traffic(kname, B, n)
        out = {
            "metric": "EKF update steps/sec + achieved HBM GB/s vs roofline, n=1000 landmarks",
            "value": corr / wall,
            "unit": "update steps/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[4], one GPU's share: {B} filters, n={n}, known association, V=2",
                       "filters_per_gpu": B, "filters_total": int(round(fsteps / K)), "landmarks": n, "state_dim": N, "corrections_per_filter_step": 2,
                       "filter_steps_per_s": fsteps / wall,
                       "ranks_seen": ranks_seen, "hbm_bytes_per_gpu": bt.device_bytes(),
                       "collectives": (backend if dist is not None else None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": f"{kname}, {krows} rows/wg",
                         "algorithmic_bytes_per_launch": st["rank2_bytes_per_launch"],
                         "avg_launch_ms": r2_avg_s * 1e3, "launches": st["rank2_launches"],
                         "rank2_share_of_step_time": st["rank2_ms"] / st["elapsed_ms"]},
            "device_elapsed_ms": st["elapsed_ms"],
        }
        if world > 1:
            out["rank_ms_per_step_min"], out["rank_ms_per_step_max"] = wall_min / K * 1e3, wall_max / K * 1e3
        if traffic is None:
            out["roofline"]["traffic_note"] = why_not
        if delayed is not None:
            out["delayed_update"] = delayed
        if callf is not None:
            out["call_fused_update"] = callf
        if active is not None:
            out["active_set_update"] = active
        if mc_main is not None:
            out["mc_consistency"] = dict(mc_main, step=W + K)
        if want_cpu:
            # bounded sample: ~10 s of CPU work incl. the untimed warm-up (each filter is 32 MB of covariance: 82 GB)
            import copy
            cfg_c = copy.copy(cfg)
            cfg_c.filters = Bc
            cfg_c.steps = 1 + W + K
            Tc = 1 + W + K
            if log is None:
                tw, li, zz, ii, _ = bt.download_log(want_truth=False)
                sub = synth.KnownLog(cfg_c, world_xy, tw[:Tc, cpu_ids], li[:Tc, cpu_ids], zz[:Tc, cpu_ids], ii[cpu_ids])
            else:
                sub = synth.KnownLog(cfg_c, log.world, log.twist[:Tc, cpu_ids], log.lm_idx[:Tc, cpu_ids],
                                     log.z_xy[:Tc, cpu_ids], log.init_xy[cpu_ids])
            out["cpu_baseline"] = cpu_baseline(sub, K, 1 + W, cores, eager_state)
            out["cpu_baseline"]["filter_ids"] = [int(cpu_ids[0]), int(cpu_ids[-1]), int(len(cpu_ids))]   # first, last, count

    # Separately reported leg: data_association() (a4/a5) over the same pool -- every robot discovers its map
    # from shuffled, unlabelled readings generated on the device; scores, gate decisions, landmark initialisation
    # and corrections all stay on the device.  Corrections are exactly confined to each filter's discovered prefix.
    if not a.no_unknown and not a.host_log:
        J, Tu = 8, 1 + W + K
        ucfg = synth.config3(steps=Tu)
        ucfg.filters, ucfg.first_filter_id, ucfg.n = B, cfg.first_filter_id, n
        uworld = synth.make_world(n, ucfg.half_extent, ucfg.min_spacing, ucfg.seed)
        bt.reset()
        bt.simulate_unknown_log(ucfg, uworld, jmax=J)
        bt.run_unknown(0, 1 + W)
        fence()
        t0 = time.perf_counter()
        su = bt.run_unknown(1 + W, Tu, time_kernels=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fence()
        uwall, ucorr, usteps = shard.reduce_throughput(t1 - t0, float(su["corrections"]), float(su["filter_steps"]),
                                                       device=red_dev)
        if rank == 0:
            kc = bt.known_counts()
            out["unknown_association"] = {
                "value": usteps / uwall, "unit": "filter steps/s",
                "corrections_per_s": ucorr / uwall, "measurement_slots": su["rank2_launches"],
                "known_landmarks_min": int(kc.min()), "known_landmarks_max": int(kc.max()),
                "active_dimension_max": 3 + 2 * int(kc.max()),
                "rank2_share_of_time": su["rank2_ms"] / su["elapsed_ms"],
                "mc_consistency": bt.mc_stats(Tu - 1)}
    # The LARGE-prefix regime of the same path: every robot explores a map it has already surveyed (phase A, untimed:
    # a host-written known-association log corrects each of the n landmarks once from the origin), then runs unknown
    # association against all n of them -- every reading is scored against 1000 landmarks (one per wavefront,
    # ekf_slam.cpp:300-309) and the winner corrected at full width (:331-390): four launches per measurement slot.
    if not a.no_unknown and not a.host_log:
        Bl, J, Tu = min(B, 512), 8, 1 + W + K
        kdl = 32                                  # delayed variant: corrections per flush ...
        Kdl = -(-K // (kdl // J)) * (kdl // J)    # ... and its step count: whole flushes (kdl / J steps each)
        lb = capi.BatchEKF(Bl, n, device=local)
        lworld = synth.make_world(n, 12.0, 0.6, 3)
        rng = np.random.default_rng(1000 + rank)
        vm = 64
        Ta = 1 + (n + vm - 1) // vm
        a_lm = np.full((Ta, Bl, vm), -1, dtype=np.int32)
        a_z = np.zeros((Ta, Bl, vm, 2))
        for t in range(1, Ta):
            idx = np.arange((t - 1) * vm, min(n, t * vm))
            a_lm[t, :, :len(idx)] = idx
            a_z[t, :, :len(idx)] = lworld[idx][None] + rng.normal(0.0, 0.005, size=(Bl, len(idx), 2))
        a_init = (lworld[None] + rng.normal(0.0, 0.005, size=(Bl, n, 2))).reshape(Bl, 2 * n)
        lb.set_tuning(a.rows, a.nt)
        lb.upload_known_log(np.zeros((Ta, Bl, 2)), a_lm, a_z, a_init)  # robot at rest at the origin: robot frame = world frame
        lb.set_call_fused(True)   # (the survey is untimed: 8 corrections per pass over Sigma)
        lb.run_known(0, Ta)
        lb.set_call_fused(False)
        lb.set_known_counts(n)
        lcfg = synth.config3(steps=Tu + 2 * Kdl)
        lcfg.filters, lcfg.first_filter_id, lcfg.n = Bl, rank * Bl, n
        lb.simulate_unknown_log(lcfg, lworld, jmax=J)
        lb.run_unknown(0, 1 + W)
        fence()
        t0 = time.perf_counter()
        sl = lb.run_unknown(1 + W, Tu, time_kernels=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fence()
        lwall, lcorr, lsteps = shard.reduce_throughput(t1 - t0, float(sl["corrections"]), float(sl["filter_steps"]),
                                                       device=red_dev)
        if rank == 0:
            kc = lb.known_counts()
            dec = lb.decisions()[1 + W:Tu]
            nmeas = int((dec > -2).sum()) * world
            r2_s = sl["rank2_ms"] / max(sl["rank2_launches"], 1) * 1e-3
            out["unknown_association_large_prefix"] = {
                "value": lsteps / lwall, "unit": "filter steps/s",
                "measurements_per_s": nmeas / lwall, "scores_per_s": nmeas * float(n) / lwall,
                "corrections_per_s": lcorr / lwall, "filters_per_gpu": Bl,
                "known_landmarks_min": int(kc.min()), "known_landmarks_max": int(kc.max()),
                "steps_timed": sl["rank2_launches"], "covariance_pass_avg_ms": r2_s * 1e3,
                "covariance_pass_share_of_time": sl["rank2_ms"] / sl["elapsed_ms"],
                "covariance_GBps": Bl * 16.0 * N * N / r2_s / 1e9,
                "frac_of_8TBps_in_the_pass": Bl * 16.0 * N * N / r2_s / 1e9 / HBM_PEAK_GBS,
                "frac_of_8TBps_end_to_end": lsteps / lwall * 16.0 * N * N / 1e9 / HBM_PEAK_GBS,
                "mc_consistency": dict(lb.mc_stats(Tu - 1), step=Tu - 1)}
        # The same leg in DELAYED mode (ekf_batch_set_update_mode): the pairs of a step stay pending across steps, every
        # reading is scored and corrected against the stored covariance minus all pending pairs, Sigma is rewritten once per
        # kdl / J steps.  Timed over the steps that follow the eager leg's; parity against an eager side pool of the first
        # filters (same global ids -> same survey noise and readings) that runs the whole sequence eagerly.
        lb.set_update_mode(kdl)
        fence()
        t0 = time.perf_counter()
        sd2 = lb.run_unknown(Tu, Tu + Kdl, time_kernels=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fence()
        dwall2, dcorr2, dsteps2 = shard.reduce_throughput(t1 - t0, float(sd2["corrections"]), float(sd2["filter_steps"]),
                                                          device=red_dev)
        ref = None
        if rank == 0:
            nref = min(Bl, 8)
            ref = capi.BatchEKF(nref, n, device=local)
            ref.upload_known_log(np.zeros((Ta, nref, 2)), a_lm[:, :nref], a_z[:, :nref], a_init[:nref])
            ref.run_known(0, Ta)
            ref.set_known_counts(n)
            import copy
            rcfg = copy.copy(lcfg)
            rcfg.filters = nref
            ref.simulate_unknown_log(rcfg, lworld, jmax=J)
            ref.run_unknown(0, Tu + Kdl)
            dec_d, dec_e = lb.decisions()[:, :nref], ref.decisions()
            sdiff = max(float(np.abs(lb.state(b) - ref.state(b)).max()) for b in range(nref))
            c_d, c_e = lb.cov(0), ref.cov(0)
            out["unknown_association_large_prefix"]["delayed"] = {
                "value": dsteps2 / dwall2, "unit": "filter steps/s", "corrections_per_flush": kdl, "steps": Kdl,
                "steps_per_flush": kdl // J, "flushes": sd2["rank2_launches"],
                "flush_avg_ms": sd2["rank2_ms"] / max(sd2["rank2_launches"], 1),
                "corrections_per_s": dcorr2 / dwall2, "speedup_vs_eager": (dsteps2 / dwall2) / (lsteps / lwall),
                "parity_filters": nref, "decisions_identical_to_eager": bool(np.array_equal(dec_d, dec_e)),
                "max_abs_state_diff_vs_eager": sdiff,
                "max_rel_cov_diff_vs_eager": float(np.abs(c_d - c_e).max() / np.abs(c_e).max()),
                "mc_consistency": dict(lb.mc_stats(Tu + Kdl - 1), step=Tu + Kdl - 1)}
        # ... and with the opt-in symmetric option (row-only reconstruction in the step kernel, mirrored flush), over the
        # steps that follow; same side pool
        lb.set_update_mode(kdl, symmetric_gather=True)
        fence()
        t0 = time.perf_counter()
        sd3 = lb.run_unknown(Tu + Kdl, Tu + 2 * Kdl, time_kernels=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fence()
        dwall3, _, dsteps3 = shard.reduce_throughput(t1 - t0, float(sd3["corrections"]), float(sd3["filter_steps"]), device=red_dev)
        if rank == 0:
            ref.run_unknown(Tu + Kdl, Tu + 2 * Kdl)
            c_d, c_e = lb.cov(0), ref.cov(0)
            out["unknown_association_large_prefix"]["delayed"]["symmetric"] = {
                "value": dsteps3 / dwall3, "unit": "filter steps/s", "steps": Kdl,
                "flush_avg_ms": sd3["rank2_ms"] / max(sd3["rank2_launches"], 1),
                "flush_form": "k_flush_sym" if lb.form_counts()["flush_mirrored"] else "full",
                "decisions_identical_to_eager": bool(np.array_equal(lb.decisions()[:, :nref], ref.decisions())),
                "max_abs_state_diff_vs_eager": max(float(np.abs(lb.state(b) - ref.state(b)).max()) for b in range(nref)),
                "max_rel_cov_diff_vs_eager": float(np.abs(c_d - c_e).max() / np.abs(c_e).max())}
            ref.close()
        lb.close()
    # The reference's own operating point at Monte-Carlo scale: configs[0] (n = 20, 1000 steps) for 8192 robots per
    # GPU, inputs simulated on the device, the whole run ONE launch with every covariance resident in LDS.
    if not a.no_small and not a.host_log:
        Bs, Ts, ns = 8192, 1000, 20
        scfg = synth.config1(steps=Ts)
        scfg.filters, scfg.first_filter_id = Bs, rank * Bs
        sworld = synth.make_world(ns, scfg.half_extent, scfg.min_spacing, scfg.seed)
        sb = capi.BatchEKF(Bs, ns, device=local)
        sb.simulate_known_log(scfg, sworld, vmax=ns)
        sb.run_known(0, 1)  # the first call initialises the map (ekf_slam.cpp:113-128)
        fence()
        t0 = time.perf_counter()
        ss = sb.run_known(1, Ts, time_kernels=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fence()
        swall, scorr, ssteps = shard.reduce_throughput(t1 - t0, float(ss["corrections"]), float(ss["filter_steps"]),
                                                       device=red_dev)
        if rank == 0:
            out["small_map_monte_carlo"] = {
                "value": ssteps / swall, "unit": "filter steps/s",
                "corrections_per_s": scorr / swall, "filters_per_gpu": Bs, "landmarks": ns, "steps": Ts - 1,
                "launches": ss["rank2_launches"], "mc_consistency": sb.mc_stats(Ts - 1)}
        sb.close()
    bt.close()
    # The other BASELINE.json configurations (N = 1 only: they are single-GPU, single-filter workloads).
    if world == 1 and not a.no_configs:
        torch.cuda.empty_cache()
        for key, leg in (("configs_1", leg_configs_1), ("configs_2", leg_configs_2), ("configs_3", leg_configs_3)):
            try:
                out[key] = leg(local, cores, want_cpu=want_cpu)
            except Exception as e:  # a failing side leg must not take the contract line down with it
                out[key] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        line = json.dumps(compact(out), separators=(",", ":"))
        # the driver keeps an 8 KB tail of the run's output: every leg must be readable there (README: what each key means)
        assert len(line) < LINE_BUDGET, f"bench line is {len(line)} characters, budget {LINE_BUDGET}"
        print(line, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
