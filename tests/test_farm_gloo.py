"""CPU tier: the multi-GPU farm logic (round-robin partition, in-flight pipelines, barriers, MAX/SUM reductions, digest
all-gather and cross-check) under torch.distributed with the gloo backend, world_size 2, with a fake matcher -- the same
adcensus_amd/farm.py code that bench.py drives with the HIP matcher over RCCL."""
import os
import time

import pytest


class FakeMatcher:
    """Stands in for the HIP pipelines: 'computes' a pair by sleeping; the output is a pure function of the pair id
    (optionally corrupted for one pair on one rank, to prove that the cross-check catches it)."""

    def __init__(self, rank, slow=0.004, corrupt=None):
        self.rank, self.slow, self.corrupt = rank, slow, corrupt
        self.inflight = {}
        self.order = []
        self.max_in_flight = 0

    def submit(self, slot, pid):
        assert slot not in self.inflight, "slot reused before it was collected"
        self.inflight[slot] = pid
        self.max_in_flight = max(self.max_in_flight, len(self.inflight))

    def wait(self, slot):
        pid = self.inflight.pop(slot)
        time.sleep(self.slow * (self.rank + 1))  # rank 1 is twice as slow
        self.order.append(pid)

    def output(self, pid):
        bad = self.corrupt is not None and self.corrupt == (self.rank, pid)
        return ("disparity map of pair %d%s" % (pid, " (bit flip)" if bad else "")).encode()


def _worker(rank, world, port, q, corrupt):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from adcensus_amd import farm
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    steps, warmup, inflight = 5, 1, 2
    batch = steps * world
    mine = farm.partition(batch, world, rank)
    fm = FakeMatcher(rank, corrupt=corrupt)

    def run_steps(n):
        farm.run_pairs([mine[i % len(mine)] for i in range(n)], fm.submit, fm.wait, inflight)
    elapsed, total = farm.timed_region(run_steps, steps=steps, warmup=warmup, dist=dist)
    timed_order = fm.order[warmup:]
    primary = {pid: farm.digest(fm.output(pid)) for pid in mine}
    recheck = {}
    for pid in farm.neighbour_pairs(batch, world, rank):
        fm.submit(0, pid)
        fm.wait(0)
        recheck[pid] = farm.digest(fm.output(pid))
    done = farm.done_counter(len(primary), dist)
    report = farm.cross_check(farm.gather_digests(primary, dist), farm.gather_digests(recheck, dist),
                              reference={str(i): farm.digest(("disparity map of pair %d" % i).encode()) for i in range(batch)})
    q.put((rank, elapsed, total, timed_order, fm.max_in_flight, done, report, mine))
    dist.destroy_process_group()


def _run(corrupt=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, corrupt)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_farm_two_ranks_gloo():
    (r0, e0, t0, o0, f0, d0, rep0, m0), (r1, e1, t1, o1, f1, d1, rep1, m1) = _run()
    assert abs(e0 - e1) < 1e-9            # MAX over ranks: identical on every rank
    assert e0 >= 5 * 0.008 * 0.9          # the slow rank's 5 pairs bound the job
    assert t0 == t1 == 10                 # SUM of steps over ranks
    assert o0 == m0 and o1 == m1          # every rank processed exactly its partition, in order, once
    assert sorted(m0 + m1) == list(range(10)) and not set(m0) & set(m1)
    assert f0 == f1 == 2                  # two pipelines in flight
    assert d0 == d1 == 10                 # completion counter == batch
    for rep in (rep0, rep1):
        assert rep["pairs"] == 10 and rep["cross_checked"] == 10 and rep["reference_checked"] == 10
        assert not rep["duplicates"] and not rep["mismatches"] and not rep["reference_mismatches"]


def test_farm_cross_check_catches_a_wrong_output():
    res = _run(corrupt=(1, 3))  # rank 1 delivers a corrupted map for pair 3 (its own pair: 3 % 2 == 1)
    for r in res:
        rep = r[6]
        assert rep["mismatches"] == [3] and rep["reference_mismatches"] == [3]


def test_partition_properties():
    from adcensus_amd import farm
    for n in (0, 1, 7, 64):
        for world in (1, 2, 4, 8):
            parts = [farm.partition(n, world, r) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            assert sorted(sum((farm.neighbour_pairs(n, world, r) for r in range(world)), [])) == list(range(n))


def test_run_pairs_keeps_the_pipelines_full():
    from adcensus_amd import farm
    for inflight in (1, 2, 3):
        fm = FakeMatcher(0, slow=0.0)
        ids = list(range(7))
        assert farm.run_pairs(ids, fm.submit, fm.wait, inflight) == 7
        assert fm.order == ids and not fm.inflight and fm.max_in_flight == min(inflight, 7)
    assert farm.run_pairs([], lambda s, p: None, lambda s: None, 2) == 0
