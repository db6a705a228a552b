"""The N>1 path on CPU: two gloo ranks exercise the shard bookkeeping bench.py uses (contiguous
batch slices, barrier, MAX-over-ranks time, whole-job throughput) and the shard invariance of the
counter-based RNG indexing (rank r's draws == slice r of the single-process draws)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multiagent_particle_envs_amd import sharding
from oracle import philox


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = 1000
    off, cnt = sharding.shard_range(B, rank, world)
    sharding.barrier("cpu")
    secs = 0.5 + 0.25 * rank               # rank 1 is the slow one
    value = sharding.whole_job_throughput(cnt * 10, secs)
    tmax = sharding.reduce_max(secs)
    pos = philox.reset_positions(1234, cnt, 3, 3, 3, 1.0, world_offset=off)
    ids = philox.action_ids(99, cnt, 7, 3, world_offset=off)
    out.put((rank, off, cnt, value, tmax, pos, ids))
    sharding.barrier("cpu")
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, off0, cnt0, v0, t0, pos0, ids0), (r1, off1, cnt1, v1, t1, pos1, ids1) = res
    assert (off0, cnt0, off1, cnt1) == (0, 500, 500, 500)
    assert t0 == t1 == 0.75                                  # MAX over ranks
    assert v0 == v1 == (500 * 10 + 500 * 10) / 0.75          # whole-job units / slowest rank
    full_pos = philox.reset_positions(1234, 1000, 3, 3, 3, 1.0)
    full_ids = philox.action_ids(99, 1000, 7, 3)
    assert np.array_equal(np.concatenate([pos0, pos1]), full_pos)
    assert np.array_equal(np.concatenate([ids0, ids1], axis=1), full_ids)


def test_shard_range_partitions():
    for B in (1, 7, 64, 65536, 524288, 1000003):
        for G in (1, 2, 3, 4, 8):
            spans = [sharding.shard_range(B, r, G) for r in range(G)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == B
            for (o, c), (o2, _) in zip(spans, spans[1:]):
                assert o + c == o2
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
