#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: EKF update steps/s + achieved HBM GB/s vs roofline, n = 1000.

Workload (config.workload): one GPU's share of BASELINE.json configs[4] -- B independent rigid2d::EKF_SLAM
filters (Monte-Carlo batch), n = 1000 landmarks (N = 2003, fp64), known association, exactly V = 2 landmark
corrections per filter step.  A bench "step" = for every filter: prediction(twist) + measurement(2 readings).
An "EKF update step" (the unit of `value`) = one landmark correction (gain + state + covariance update,
ekf_slam.cpp:137-192); filter steps/s are reported next to it.  Inputs (twists, readings) are uploaded to HBM
before the timed region.  Multi-GPU: one process per GPU, filters sharded by global id, no data-path
collective; RCCL carries only the final throughput reduction ("scaling": "weak").

  python bench.py [--gpus N] [--steps K] [--warmup W] [--filters B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--filters", type=int, default=0, help="filters per GPU (0 = 4096, reduced to fit HBM)")
    ap.add_argument("--landmarks", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-filters", type=int, default=0)
    ap.add_argument("--delayed-k", type=int, default=32,
                    help="also time the delayed rank-2k update with this many corrections per flush (0 = skip)")
    ap.add_argument("--no-active-set", action="store_true", help="skip the active-set leg")
    ap.add_argument("--no-unknown", action="store_true", help="skip the batched unknown-association leg")
    ap.add_argument("--no-small", action="store_true", help="skip the small-map (n = 20) Monte-Carlo leg")
    ap.add_argument("--host-log", action="store_true",
                    help="generate the synthetic log on the host (numpy) and upload it, instead of on the device")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--nt", type=int, default=-1)
    return ap.parse_args()


def cpu_baseline(sub, K, t_warm, cores, gpu_state):
    """The CPU checker's structured restatement ("port"), OpenMP over filters, on a bounded sample of the
    same log (its first filters), timed on this box's host cores.  Doubles as a parity spot-check."""
    import numpy as np
    from oracle import binding as ob  # checker / baseline only -- never on the product path
    st, _, stats = ob.batch_run_known(sub, ob.STRUCTURED, t_warm=t_warm, nthreads=cores, want_cov=False, fast=True)
    B = sub.twist.shape[1]
    return {"value": stats["corrections"] / stats["seconds"], "unit": "update steps/s", "cores": stats["threads"],
            "kind": "port",
            "sample": f"{B} filters x {K} timed steps x 2 corrections at n={sub.cfg.n} "
                      f"({stats['corrections']} corrections, {stats['seconds']:.1f} s): structured O(N^2) C "
                      f"restatement (oracle/ekf_oracle.c mode 1, gcc -O3 -mavx2 -mfma, OpenMP over filters); the "
                      f"reference's own dense Armadillo path is O(N^3) per correction and cannot be built here",
            "max_abs_state_diff_vs_gpu": float(np.abs(st - gpu_state).max())}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from ekf_slam_ml_amd import capi, synth

    n = a.landmarks
    N = 3 + 2 * n
    ld = (N + 15) // 16 * 16
    per_filter = N * ld * 8 + 6 * ld * 8 + 4 * n * 8
    free, total = torch.cuda.mem_get_info(local)
    B = a.filters if a.filters > 0 else 4096
    B = max(1, min(B, int(0.90 * free // per_filter), 65535))
    K, W = a.steps, a.warmup
    T = W + K + 1  # step 0 = first measurement() call (landmark initialisation, no corrections)

    from ekf_slam_ml_amd import shard
    first_id, count = shard.shard(B * world, world, rank)  # weak scaling: B filters per GPU, global ids
    assert count == B
    cfg = synth.config5(filters=B, steps=T, first_filter_id=first_id, n=n)
    bt = capi.BatchEKF(B, n, device=local)
    bt.set_tuning(a.rows, a.nt)
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
    # the only collective of the job: RCCL all-reduce of three scalars (max wall, summed work)
    wall, corr, fsteps = shard.reduce_throughput(wall, corr, fsteps, device=red_dev)
    eager_state = [bt.state(b) for b in range(min(B, 4))] if rank == 0 else None

    # Second, separately reported leg: the SAME K steps with the delayed
    # rank-2k covariance update.  Its traffic is different by construction, so it has its own declared
    # bytes and never enters `value` / `roofline` above.
    delayed = None
    if a.delayed_k > 0:
        bt.reset()
        bt.set_update_mode(a.delayed_k)
        bt.run_known(0, 1 + W)
        fence()
        t0 = time.perf_counter()
        sd = bt.run_known(1 + W, 1 + W + K, time_kernels=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fence()
        dwall, dcorr, _ = shard.reduce_throughput(t1 - t0, float(sd["corrections"]), float(sd["filter_steps"]),
                                                  device=red_dev)
        if rank == 0:
            kk = float(a.delayed_k)
            Nf = float(N)
            # declared algorithmic bytes per correction: flush share (Sigma read+write + factor reads) +
            # average factor read of the gain step + base gathers + factor/state append
            per_corr = (16.0 * Nf * Nf + 32.0 * Nf * kk) / kk + 16.0 * Nf * (kk - 1.0) + 14.0 * 8.0 * Nf
            dstate = [bt.state(b) for b in range(min(B, 4))]
            delayed = {"value": dcorr / dwall, "unit": "update steps/s", "corrections_per_flush": a.delayed_k,
                       "ms_per_step": dwall / K * 1e3, "flushes": sd["rank2_launches"],
                       "flush_avg_ms": sd["rank2_ms"] / max(sd["rank2_launches"], 1),
                       "flush_share_of_time": sd["rank2_ms"] / sd["elapsed_ms"],
                       "declared_bytes_per_correction": per_corr,
                       "achieved_GBps_on_declared_bytes": dcorr / world * per_corr / dwall / 1e9,
                       "frac_of_8TBps": dcorr / world * per_corr / dwall / 1e9 / HBM_PEAK_GBS,
                       "speedup_vs_eager": (dcorr / dwall) / (corr / wall),
                       "max_abs_state_diff_vs_eager": float(max(np.abs(x - y).max() for x, y in zip(dstate, eager_state))),
                       "note": "Sigma = Sigma_base - sum K_j (H Sigma)_j kept as factors, rewritten once per "
                               "k corrections; results equal the eager path to rounding (tests/test_gpu_delayed.py)"}
        bt.set_update_mode(0)

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
            astate = [bt.state(b) for b in range(min(B, 4))]
            tch = bt.touched()
            active = {"value": acorr / awall, "unit": "update steps/s", "ms_per_step": awall / K * 1e3,
                      "touched_landmarks_mean": float(tch.mean()), "touched_landmarks_max": int(tch.max()),
                      "declared_bytes_per_correction": 16.0 * N * (3 + 2 * float(tch.mean())),
                      "rank2_share_of_time": sa["rank2_ms"] / sa["elapsed_ms"],
                      "speedup_vs_eager": (acorr / awall) / (corr / wall),
                      "bit_identical_to_eager": bool(all(np.array_equal(x, y) for x, y in zip(astate, eager_state))),
                      "note": "workload-dependent: rows of never-corrected landmarks have K = 0 exactly and are "
                              "skipped (the 2 nearest landmarks of a slowly moving robot stay the same for many "
                              "steps, so few are ever touched here); with every landmark corrected it degenerates "
                              "to the dense stream"}
        bt.set_active_set(False)

    out = None
    if rank == 0:
        r2_avg_s = st["rank2_ms"] / max(st["rank2_launches"], 1) * 1e-3
        achieved = st["rank2_bytes_per_launch"] / r2_avg_s / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "rank2_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("filters") == B and tj.get("n") == n:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "EKF update steps/sec + achieved HBM GB/s vs roofline, n=1000 landmarks",
            "value": corr / wall,
            "unit": "update steps/s (1 update step = 1 landmark correction: gain + state + covariance)",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[4], one GPU's share: {B} independent EKF_SLAM filters per "
                                   f"GPU, n={n} landmarks (N={N}), known association, V=2 corrections per filter step",
                       "filters_per_gpu": B, "landmarks": n, "state_dim": N, "corrections_per_filter_step": 2,
                       "filter_steps_per_s": fsteps / wall, "sharding": f"independent filters x {world} GPUs",
                       "hbm_bytes_per_gpu": bt.device_bytes()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "ekf::k_rank2<16,true,256> (Sigma -= K*(H*Sigma), ekf_slam.cpp:191-192)",
                         "algorithmic_bytes_per_launch": st["rank2_bytes_per_launch"],
                         "avg_launch_ms": r2_avg_s * 1e3, "launches": st["rank2_launches"],
                         "rank2_share_of_step_time": st["rank2_ms"] / st["elapsed_ms"]},
            "device_elapsed_ms": st["elapsed_ms"],
        }
        if delayed is not None:
            out["delayed_update"] = delayed
        if active is not None:
            out["active_set_update"] = active
        if not a.host_log:
            # Monte-Carlo consistency of the batch against the simulated ground truth (f4)
            out["mc_consistency"] = bt.mc_stats(T - 1)
        if world == 1 and not a.no_cpu_baseline:
            # the box's CPU share for a one-GPU job is 16 cores (the machine reports all 256)
            cores = int(os.environ.get("EKF_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
            # bounded sample: ~10 s of CPU work incl. the untimed warm-up (each filter is 32 MB of covariance: 82 GB)
            Bc = a.cpu_filters if a.cpu_filters > 0 else min(B, 160 * cores)
            import copy
            cfg_c = copy.copy(cfg)
            cfg_c.filters = Bc
            if log is None:
                tw, li, zz, ii, _ = bt.download_log(want_truth=False)
                sub = synth.KnownLog(cfg_c, world_xy, tw[:, :Bc], li[:, :Bc], zz[:, :Bc], ii[:Bc])
            else:
                sub = synth.KnownLog(cfg_c, log.world, log.twist[:, :Bc], log.lm_idx[:, :Bc], log.z_xy[:, :Bc],
                                     log.init_xy[:Bc])
            gpu_state = np.stack([bt.state(b) for b in range(Bc)])  # (state after the last leg that ran)
            out["cpu_baseline"] = cpu_baseline(sub, K, 1 + W, cores, gpu_state)

    # Last, separately reported leg: data_association() (a4/a5) over the same pool -- every robot discovers its map
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
                "value": usteps / uwall, "unit": "filter steps/s (1 step = prediction + data_association of <= 8 readings)",
                "corrections_per_s": ucorr / uwall, "measurement_slots": su["rank2_launches"],
                "known_landmarks_min": int(kc.min()), "known_landmarks_max": int(kc.max()),
                "rank2_share_of_time": su["rank2_ms"] / su["elapsed_ms"],
                "mc_consistency": bt.mc_stats(Tu - 1),
                "note": "configs[2]'s world and sensor for every filter of the pool; landmarks are appended in discovery "
                        "order, so each filter's corrections stream only its leading 3 + 2*known block (bit-identical "
                        "to the full-width update, tests/test_gpu_batch_unknown.py)"}
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
                "value": ssteps / swall, "unit": "filter steps/s (1 step = prediction + measurement of the visible landmarks)",
                "corrections_per_s": scorr / swall, "filters_per_gpu": Bs, "landmarks": ns, "steps": Ts - 1,
                "launches": ss["rank2_launches"], "mc_consistency": sb.mc_stats(Ts - 1),
                "note": "BASELINE.json configs[0] (the reference's n = 20 known-association run) for every filter; "
                        "bit-identical to the per-step replay (tests/test_gpu_pool_small.py)"}
        sb.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    bt.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
