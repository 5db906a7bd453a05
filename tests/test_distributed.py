"""CPU (not gpu): the N > 1 path of bench.py -- sharding by global filter id, no data-path collective,
final reduction over torch.distributed -- rehearsed with world_size 2 on the gloo backend.  The compute
of each rank is the CPU checker here (there is no GPU in this container); what is under test is the
partition + reduction logic that the GPU ranks use unchanged with backend nccl (= RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ekf_slam_ml_amd import shard, synth

N_LM, TOTAL, STEPS = 12, 6, 6


def test_shard_is_a_partition():
    for total in (0, 1, 7, 8, 4096, 32768):
        for world in (1, 2, 3, 8):
            blocks = [shard.shard(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == total
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f1 == f0 + c0
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        shard.shard(4, 2, 2)


def _run_rank(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding as ob
        first, count = shard.shard(TOTAL, world, rank)
        log = synth.make_known_log(synth.config5(filters=count, steps=STEPS, first_filter_id=first, n=N_LM))
        st, _, stats = ob.batch_run_known(log, ob.STRUCTURED, t_warm=1, nthreads=1, fast=False)
        wall, corr, fsteps = shard.reduce_throughput(1.0 + rank, stats["corrections"], count * (STEPS - 1))
        poses = shard.gather_poses(st[:, :3])
        seen = shard.count_ranks()
        dist.barrier()
        if rank == 0:
            q.put((wall, corr, fsteps, poses, seen))
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_process():
    from oracle import binding as ob
    ob.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    wall, corr, fsteps, poses, seen = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = synth.make_known_log(synth.config5(filters=TOTAL, steps=STEPS, n=N_LM))
    st, _, stats = ob.batch_run_known(full, ob.STRUCTURED, t_warm=1, nthreads=1, fast=False)
    assert seen == 2                                     # all-reduce of ones: every rank took part
    assert wall == 2.0                                   # max over ranks
    assert corr == stats["corrections"] == TOTAL * (STEPS - 1) * 2
    assert fsteps == TOTAL * (STEPS - 1)
    assert np.array_equal(poses, st[:, :3])              # the sharded job IS the unsharded job


def test_reduction_without_process_group_is_identity():
    assert shard.reduce_throughput(1.5, 10, 5) == (1.5, 10, 5)
    assert shard.gather_poses(np.ones((2, 3))).shape == (2, 3)
    assert shard.count_ranks() == 1
