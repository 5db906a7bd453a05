"""-m gpu: bench.py's multi-rank path on the real HIP library.  One MI355X per test box, and RCCL refuses two ranks
on one device, so the ranks rendezvous over gloo (EKF_DIST_BACKEND=gloo, bench.py's rehearsal mode) and share the GPU;
everything else is the path the driver launches for --gpus N: torch.distributed.run, one process per rank, global
filter ids from shard.shard(), barrier + synchronize around the timed region, MAX over ranks of the wall time, SUM of
the work, one JSON line from rank 0.  (The sharding arithmetic itself is covered on the CPU by tests/test_distributed.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])
ARGS = ["--steps", "3", "--warmup", "1", "--filters", "96", "--landmarks", "300", "--only-main", "--no-cpu-baseline"]


def _line(out):
    rows = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(rows) == 1, out[-2000:]          # ONE JSON line, printed by rank 0 only
    return json.loads(rows[0])


def test_two_ranks_through_torchrun_aggregate_like_one():
    env = dict(os.environ, EKF_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + ARGS, capture_output=True,
                         text=True, timeout=600, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2"] + ARGS, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = _line(one.stdout), _line(two.stdout)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["scaling"] == "weak"
    assert b["config"]["ranks_seen"] == 2 and b["config"]["filters_per_gpu"] == a["config"]["filters_per_gpu"] == 96
    # weak scaling: every rank does the single-rank job's work -> twice the corrections in the aggregate
    work_a = a["value"] * a["ms_per_step"] * a["steps"]
    work_b = b["value"] * b["ms_per_step"] * b["steps"]
    assert abs(work_b / work_a - 2.0) < 1e-6   # (the line carries 9 significant digits)
    assert b["metric"] == a["metric"] and b["unit"] == a["unit"] and b["steps"] == 3 and b["warmup"] == 1
    # the driver's contract keys, on both lines
    for d in (a, b):
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in d, key
        assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64" and "workload" in d["config"]
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
        assert "traffic" in r and r["achieved"] > 0
        # value = whole-job corrections / max-over-ranks time: consistent with ms_per_step
        assert abs(d["value"] * d["ms_per_step"] * 1e-3 / (d["n_gpus"] * 96 * 2) - 1.0) < 1e-6


def test_launcher_rank_count_mismatch_is_refused():
    env = dict(os.environ, EKF_DIST_BACKEND="gloo", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"] + ARGS, capture_output=True,
                       text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "--gpus 4 but the launcher started 2" in (r.stderr + r.stdout)


def test_rccl_reductions_on_the_device():
    """The collectives of the N > 1 job on the RCCL backend itself ("nccl" on ROCm): a single-rank communicator on the
    test box's GPU runs the very calls bench.py makes at N > 1 -- fp64 MAX / SUM all-reduces on device tensors, the
    barrier, the pose all-gather."""
    code = r'''
import os, sys
sys.path.insert(0, os.environ["EKF_ROOT"])
import torch, torch.distributed as dist
from ekf_slam_ml_amd import shard
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
w, c, f = shard.reduce_throughput(1.25, 1000.0, 500.0, device="cuda")
assert (w, c, f) == (1.25, 1000.0, 500.0), (w, c, f)
assert shard.count_ranks(device="cuda") == 1
p = shard.gather_poses([[0.1, 0.2, 0.3], [1.0, 2.0, 3.0]], device="cuda")
assert p.shape == (2, 3) and p[1, 2] == 3.0
dist.destroy_process_group()
print("rccl ok")
'''
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), EKF_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout + r.stderr)[-2000:]


def _bench(extra, env=None, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                       timeout=timeout, env=env or dict(os.environ), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return _line(r.stdout)


def test_consistency_and_delayed_parity_on_the_drivers_command_line():
    """`--steps 20 --warmup 5` (the driver's arguments) rounds the delayed leg to 32 steps, so the log is longer than the
    contract leg's run: the Monte-Carlo consistency figures must still be taken at the step the pool stands at (NEES of
    the same order as for `--steps 16 --warmup 4`, where both legs run the same steps), and the delayed leg must carry its
    own parity figures at its own step count."""
    side = ["--filters", "64", "--no-active-set", "--no-unknown", "--no-small", "--no-configs", "--no-call-fused",
            "--cpu-filters", "4"]
    a = _bench(["--steps", "20", "--warmup", "5"] + side)
    b = _bench(["--steps", "16", "--warmup", "4"] + side)
    for d, last in ((a, 25), (b, 20)):
        # [nees_mean, nees_max, rmse_xy, rmse_theta, mean_trace_pose_cov, frac_nees_below_95pct, step] (bench.compact)
        mc = d["mc_consistency"]
        assert mc[6] == last
        assert 0.5 < mc[0] < 30.0 and mc[2] < 0.2 and mc[5] > 0.3, mc
    assert 1 / 3 < a["mc_consistency"][0] / b["mc_consistency"][0] < 3
    for d, kd in ((a, 32), (b, 16)):
        dl = d["delayed_update"]
        assert dl["steps"] == kd and dl["parity_step"] == d["warmup"] + kd
        assert dl["max_abs_state_diff_vs_eager"] < 1e-9 and dl["max_rel_cov_diff_vs_eager"] < 1e-9
        assert dl["max_abs_state_diff_vs_cpu_port"] < 1e-9 and dl["max_rel_cov_diff_vs_cpu_port"] < 1e-9


def test_four_ranks_with_unequal_shards():
    """4 ranks (gloo rendezvous on the one GPU of the test box) over a job of 4 * 24 + 3 filters: blocks of 25, 25, 25 and
    24 filters by global id; the aggregate counts every filter once and the line shows the per-rank spread."""
    total = 4 * 24 + 3
    env = dict(os.environ, EKF_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    args = ["--steps", "3", "--warmup", "1", "--filters", "25", "--total-filters", str(total), "--landmarks", "300",
            "--delayed-k", "0", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"),
                        "--gpus", "4"] + args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 4 and d["config"]["ranks_seen"] == 4 and d["config"]["filters_total"] == total
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / (total * 2) - 1.0) < 1e-6     # every filter counted once
    assert d["rank_ms_per_step_min"] <= d["rank_ms_per_step_max"] == d["ms_per_step"]
    # side legs are not run at N > 1
    for key in ("call_fused_update", "active_set_update", "unknown_association", "small_map_monte_carlo", "configs_1"):
        assert key not in d, key


def test_two_ranks_run_the_delayed_legs():
    """At N > 1 the job runs the contract leg plus the delayed leg and its symmetric variant: every rank takes part in their
    reductions (no rank-0-only collective), and the aggregate counts both ranks' corrections."""
    env = dict(os.environ, EKF_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    args = ["--steps", "4", "--warmup", "1", "--filters", "40", "--landmarks", "300", "--delayed-k", "4", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2"] + args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    dl = d["delayed_update"]
    assert d["n_gpus"] == 2 and dl["steps"] == 4 and dl["flushes"] == 2
    assert abs(dl["value"] * dl["ms_per_step"] * 1e-3 / (2 * 40 * 2) - 1.0) < 1e-6       # both ranks' filters, 2 corrections each
    assert dl["max_abs_state_diff_vs_eager"] < 1e-9 and dl["symmetric"]["max_abs_state_diff_vs_eager"] < 1e-9
    assert dl["symmetric"]["value"] > 0 and dl["symmetric"]["flush_form"] == "k_flush_sym"



def test_one_rank_under_the_launcher_runs_the_rccl_path():
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` with the REAL backend: bench.py's own
    init_process_group("nccl", device_id=...) (RCCL on ROCm), its barriers around the timed region and the scalar
    all-reduces of shard.py run through RCCL on the device -- the very code path of the driver's N = 2, 4, 8 jobs, which this
    one-GPU box cannot start (RCCL refuses two ranks on one device; the multi-rank tests above rendezvous over gloo).
    The line must equal the launcher-less run's in everything but time."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("EKF_DIST_BACKEND", None)
    args = ["--steps", "4", "--warmup", "1", "--filters", "96", "--landmarks", "300", "--no-cpu-baseline", "--no-active-set",
            "--no-unknown", "--no-small", "--no-configs", "--no-call-fused", "--delayed-k", "8"]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1"] + args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["config"]["collectives"] == "nccl" and d["config"]["ranks_seen"] == 1 and d["n_gpus"] == 1
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / (96 * 2) - 1.0) < 1e-6
    dl = d["delayed_update"]
    assert dl["max_abs_state_diff_vs_eager"] < 1e-9 and abs(dl["value"] * dl["ms_per_step"] * 1e-3 / (96 * 2) - 1.0) < 1e-6
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, capture_output=True,
                           text=True, timeout=900, env=env, cwd=ROOT)
    assert plain.returncode == 0, plain.stderr[-3000:]
    p = _line(plain.stdout)
    assert p["config"]["collectives"] is None
    assert p["mc_consistency"] == d["mc_consistency"]            # same filters, same inputs, same end state
    assert p["delayed_update"]["max_abs_state_diff_vs_eager"] == dl["max_abs_state_diff_vs_eager"]
