"""CPU: the N > 1 path (sharding + timing protocol of bench.py) with two gloo
ranks.  No kernels are involved: pairs are independent, the distributed part
of the hot path is exactly this protocol."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from superpoints_registration_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert sharding.rank_world() == (rank, rank, world)
    seeds = sharding.pair_seeds(rank, 3)
    work = {"n": 0}

    def step():
        work["n"] += 1
        time.sleep(0.02 * (rank + 1))          # rank 1 is the slow one

    elapsed, own = sharding.timed_steps(step, steps=4, dist=dist, return_own=True)
    per_rank = sharding.gather_ms(1e3 * own / 4, dist=dist)        # what bench.py prints as ms_per_step_per_rank
    mine = list(sharding.shard_range(11, rank, world))
    gathered = [None] * world
    dist.all_gather_object(gathered, (seeds, mine, work["n"], elapsed, per_rank))
    q.put((rank, gathered))
    dist.destroy_process_group()


def test_two_rank_protocol():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = results[0]
    (s0, m0, n0, e0, pr0), (s1, m1, n1, e1, pr1) = g
    assert pr0 == pr1 and len(pr0) == 2                # every rank sees every rank's own step time
    assert 20 * 0.9 <= pr0[0] <= 20 * 3 and pr0[1] >= 40 * 0.9 and pr0[1] > pr0[0]   # rank 1 sleeps twice as long
    assert max(pr0) * 4 / 1e3 <= e0 * 1.001            # the reported MAX covers the slowest rank
    assert not set(s0) & set(s1)                       # disjoint synthetic pairs
    assert sorted(m0 + m1) == list(range(11)) and abs(len(m0) - len(m1)) <= 1
    assert n0 == n1 == 4                               # exactly K steps each
    assert e0 == e1                                    # MAX over ranks, same on every rank
    assert e0 >= 4 * 0.04 * 0.9                        # the slow rank's time
    assert abs(sharding.throughput(3, 4, 2, e0) - 24 / e0) < 1e-9


def test_shard_range_properties():
    for n in (0, 1, 7, 64):
        for w in (1, 2, 3, 8):
            parts = [list(sharding.shard_range(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
