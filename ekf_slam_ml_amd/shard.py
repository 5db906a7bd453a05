"""Multi-GPU sharding of a Monte-Carlo batch of independent filters.

Filters share nothing (each is one rigid2d::EKF_SLAM object, ekf_slam.hpp:61-65), so the batch is
partitioned by GLOBAL filter id into contiguous blocks, one process per GPU, with NO data-path
collective.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU
tests) carries only the final reduction: max wall time, summed work counts, and optionally the
gathered poses.  A single filter does not shard ("replicas only")."""
from __future__ import annotations


def shard(total_filters: int, world: int, rank: int):
    """(first_global_id, count) of rank's contiguous block; blocks differ by at most one filter."""
    if not (0 <= rank < world) or total_filters < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(total_filters, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def reduce_throughput(wall_s: float, corrections: float, filter_steps: float, device="cpu"):
    """Whole-job figures: (max wall over ranks, sum corrections, sum filter steps)."""
    dist = _dist()
    if dist is None:
        return wall_s, corrections, filter_steps
    import torch
    tmax = torch.tensor([wall_s], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    sums = torch.tensor([corrections, filter_steps], dtype=torch.float64, device=device)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return float(tmax.item()), float(sums[0].item()), float(sums[1].item())


def wall_spread(wall_s: float, device="cpu"):
    """(min, max) over ranks of a rank-local wall time: a straggler shows up as a gap between the two."""
    dist = _dist()
    if dist is None:
        return wall_s, wall_s
    import torch
    t = torch.tensor([wall_s, -wall_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(-t[1].item()), float(t[0].item())


def count_ranks(device="cpu"):
    """Number of ranks that took part in the job, by an all-reduce of ones (1 without a process group): the bench line
    reports it so that a multi-GPU figure can be told from one rank's."""
    dist = _dist()
    if dist is None:
        return 1
    import torch
    one = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(round(float(one.item())))


def gather_poses(poses, device="cpu"):
    """All ranks' [B_r, 3] pose blocks concatenated in global filter order (equal B_r required)."""
    dist = _dist()
    import torch
    t = torch.as_tensor(poses, dtype=torch.float64).to(device).contiguous()
    if dist is None:
        return t.cpu().numpy()
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out).cpu().numpy()
