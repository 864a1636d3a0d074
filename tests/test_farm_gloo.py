"""CPU tier: the multi-GPU farm logic (static partition, barriers, MAX/SUM reductions) under
torch.distributed with the gloo backend, world_size 2, with a fake step runner."""
import os
import time

import pytest


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from adcensus_amd import farm
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    done = []

    def run_steps(n):
        for _ in range(n):
            time.sleep(0.01 * (rank + 1))  # rank 1 is twice as slow
            done.append(1)
    elapsed, total = farm.timed_region(run_steps, steps=5, warmup=1, dist=dist)
    mine = farm.partition(11, world, rank)
    q.put((rank, elapsed, total, len(done), mine))
    dist.destroy_process_group()


def test_farm_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, e0, t0, d0, m0), (r1, e1, t1, d1, m1) = res
    assert abs(e0 - e1) < 1e-9            # MAX over ranks: identical on every rank
    assert e0 >= 5 * 0.02 * 0.9           # the slow rank's 5 steps bound the job
    assert t0 == t1 == 10                 # SUM of steps over ranks
    assert d0 == d1 == 6                  # warmup + steps executed locally
    assert sorted(m0 + m1) == list(range(11)) and not set(m0) & set(m1)  # every pair exactly once


def test_partition_properties():
    from adcensus_amd import farm
    for n in (0, 1, 7, 64):
        for world in (1, 2, 4, 8):
            parts = [farm.partition(n, world, r) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
