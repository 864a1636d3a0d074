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


# ------------------------------------------------------------------------------------------------ pull queue (configs[4])
class FailingMatcher(FakeMatcher):
    """Raises MatchFailed on the k-th wait of this rank (a GPU that dies mid-batch)."""

    def __init__(self, rank, fail_after=None, **kw):
        super().__init__(rank, **kw)
        self.fail_after, self.waits = fail_after, 0

    def wait(self, slot):
        from adcensus_amd import farm
        self.waits += 1
        if self.fail_after is not None and self.waits > self.fail_after:
            self.inflight.pop(slot)
            raise farm.MatchFailed("simulated device loss on rank %d" % self.rank)
        super().wait(slot)


def test_pull_queue_threads_local_store():
    """Four 'ranks' (threads) on one LocalStore: every pair of a batch of 64 is done exactly once, also when one of them
    fails after three pairs with two in flight (its pairs go back into the queue)."""
    import threading
    from adcensus_amd import farm
    for fail_rank in (None, 2):
        store = farm.LocalStore()
        res = {}

        def worker(r):
            q = farm.PullQueue(store, 64, world=4)
            fm = FailingMatcher(r, fail_after=3 if r == fail_rank else None, slow=0.0005)
            res[r] = farm.run_queue(q, fm.submit, fm.wait, inflight=2) + (fm.max_in_flight,)
        ts = [threading.Thread(target=worker, args=(r,)) for r in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(60)
        done = sorted(sum((res[r][0] for r in range(4)), []))
        assert done == list(range(64)), done
        assert [res[r][1] for r in range(4)] == [r == fail_rank for r in range(4)]
        if fail_rank is not None:
            assert len(res[fail_rank][0]) == 3
        assert all(res[r][2] == 2 for r in range(4))


def test_pull_queue_requeue_and_all_failed():
    from adcensus_amd import farm
    store = farm.LocalStore()
    q = farm.PullQueue(store, 8, world=2)
    bad = FailingMatcher(0, fail_after=2, slow=0.0)
    mine, failed = farm.run_queue(q, bad.submit, bad.wait, inflight=2)
    assert failed and mine == [0, 1]  # pairs 2 and 3 were in flight: back in the queue
    good = FakeMatcher(1, slow=0.0)
    mine2, failed2 = farm.run_queue(farm.PullQueue(store, 8, world=2), good.submit, good.wait, inflight=2)
    assert not failed2 and sorted(mine2) == [2, 3, 4, 5, 6, 7]
    # every rank retired with pairs unfinished: an error, not a silent short batch
    store2 = farm.LocalStore()
    q2 = farm.PullQueue(store2, 3, world=1)
    worse = FailingMatcher(0, fail_after=0, slow=0.0)
    assert farm.run_queue(q2, worse.submit, worse.wait, inflight=1) == ([], True)
    q3 = farm.PullQueue(store2, 3, world=1)
    store2.add("next", 100)  # (nothing left in the main counter; the re-queued pair is claimed and fails again)
    assert farm.run_queue(q3, worse.submit, worse.wait, inflight=1) == ([], True)
    store2.set("rq_head", str(store2.add("rq_tail", 0)))  # nothing claimable any more
    with pytest.raises(RuntimeError):
        farm.run_queue(farm.PullQueue(store2, 3, world=1), good.submit, good.wait, inflight=1)


def test_pull_queue_claim_is_unambiguous():
    """Advisor finding (round 3): compare_set returns the CURRENT value also when it fails, so a claim that writes a bare
    "head + 1" looks successful to a rank that lost the race to exactly one other rank -- two ranks would compute the same
    re-queued pair.  The claim now carries a claimant token.  Model: both ranks have read head = 0; the slower one's
    compare_set runs after the faster one's."""
    from adcensus_amd import farm

    class RacyStore(farm.LocalStore):
        """Lets a second queue claim the entry between this queue's read of rq_head and its compare_set."""
        def __init__(self):
            super().__init__()
            self.intruder = None

        def compare_set(self, key, expected, desired):
            if key == "rq_head" and self.intruder is not None:
                q, self.intruder = self.intruder, None
                self.stolen = q.try_pull()  # the other rank wins the race
            return super().compare_set(key, expected, desired)

    store = RacyStore()
    a, b = farm.PullQueue(store, 0, world=2), farm.PullQueue(store, 0, world=2)
    a.requeue(7)
    store.intruder = b
    got_a = a.try_pull()   # a read head = 0, then b claimed entry 1, then a's compare_set failed
    assert store.stolen == 7 and got_a is None
    a.requeue(9)
    assert a.try_pull() == 9 and b.try_pull() is None


def _queue_worker(rank, world, port, q, batch, fail_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from adcensus_amd import farm
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    store = farm.job_store(dist, "batch0")
    fm = FailingMatcher(rank, fail_after=2 if rank == fail_rank else None, slow=0.002)
    queue = farm.PullQueue(store, batch, world=world)

    def run(_n):
        run.result = farm.run_queue(queue, fm.submit, fm.wait, inflight=2)
    elapsed, _ = farm.timed_region(run, steps=batch, warmup=0, dist=dist)
    mine, failed = run.result
    primary = {pid: farm.digest(fm.output(pid)) for pid in mine}
    all_primary = farm.gather_digests(primary, dist)
    recheck = {}
    if not failed:
        for pid in farm.recheck_assignment(all_primary, rank):
            recheck[pid] = farm.digest(fm.output(pid))
    done = farm.done_counter(len(primary), dist)
    report = farm.cross_check(all_primary, farm.gather_digests(recheck, dist))
    q.put((rank, elapsed, sorted(mine), failed, done, report))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [None, 1])
def test_pull_queue_four_ranks_gloo_batch_64(fail_rank):
    """BASELINE.json configs[4] in the CPU tier: a fixed batch of 64 pairs pulled by 4 ranks over the job's TCP store
    (gloo for the barriers / reductions / digest all-gather); with one rank failing after two pairs the other three finish
    the batch, every pair exactly once, and the slow ranks take fewer pairs than the fast one."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000) + (0 if fail_rank is None else 7)
    procs = [ctx.Process(target=_queue_worker, args=(r, 4, port, q, 64, fail_rank)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(sum((r[2] for r in res), [])) == list(range(64))
    assert len({r[1] for r in res}) == 1            # MAX-reduced elapsed time: the same on every rank
    assert all(r[4] == 64 for r in res)             # completion counter == batch
    assert [r[3] for r in res] == [r[0] == fail_rank for r in res]
    if fail_rank is None:
        assert len(res[0][2]) > len(res[3][2])      # rank 0 is 4x faster than rank 3 (FakeMatcher): it pulled more pairs
    else:
        assert len(res[fail_rank][2]) == 2
    for r in res:
        assert r[5]["pairs"] == 64 and not r[5]["duplicates"] and not r[5]["mismatches"]
        assert r[5]["cross_checked"] >= 64 - 2 - 16  # everything a healthy rank delivered was recomputed by another healthy rank


def test_pull_queue_eight_ranks_gloo_batch_64():
    """The largest rank count of BASELINE.json configs[4] (a fixed batch of 64 pairs over 1 / 2 / 4 / 8 GPUs) in the CPU tier:
    8 ranks pull the batch from the job's TCP store; every pair exactly once, one elapsed time, the completion counter equals
    the batch, the digests of every pair agree with their recomputation on another rank, faster ranks take more pairs."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_queue_worker, args=(r, 8, port, q, 64, None)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(sum((r[2] for r in res), [])) == list(range(64))
    assert len({r[1] for r in res}) == 1
    assert all(r[4] == 64 for r in res) and not any(r[3] for r in res)
    assert len(res[0][2]) > len(res[7][2])
    for r in res:
        assert r[5]["pairs"] == 64 and not r[5]["duplicates"] and not r[5]["mismatches"] and r[5]["cross_checked"] == 64
