"""Pair farm: independent stereo pairs are the unit of multi-GPU work (BASELINE.json configs[4]).

One process per GPU (torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests).
There is NO data-path collective: every Match is independent.  The only communication is
  * a completion barrier before / after the timed region,
  * a MAX all-reduce of the per-rank elapsed time (the job is as slow as its slowest GPU),
  * a SUM all-reduce of the processed-pair counter (the "trivial completion barrier" of the north star), and
  * an all-gather of {pair id: SHA-256 of its disparity map} -- a few dozen bytes per pair -- so that every output can be
    compared with the output of the SAME pair computed by another GPU (and with the committed 1-GPU table).
This module holds exactly that logic, independent of HIP, so that bench.py (HIP matcher, RCCL) and the CPU test tier
(fake matcher, gloo, world_size 2) run the same code.

Vocabulary: a *batch* is a list of pair ids 0..B-1 (pair i is the synthetic pair of seed 12345+i); a *matcher* owns
`inflight` pipelines (ADCensusStereo objects on separate streams) and exposes submit(slot, pair_id) / wait(slot).

Two ways to hand the batch out:
  * static   partition(): pair i -> rank i % world (weak-scaling runs: every GPU does exactly `steps` pairs);
  * pull     PullQueue: every rank pulls the next pair index from ONE shared counter (an atomic add on the job's
             rendezvous store -- a few bytes over TCP, no collective), so a slow GPU simply pulls fewer pairs
             (BASELINE.json configs[4]: a FIXED batch of 64 pairs over 1/2/4/8 GPUs).  A pair whose Match fails is put
             back (requeue) and the failing rank retires; the others finish the batch ("failure detection").
"""
import hashlib
import time


def partition(n_items, world, rank):
    """Static round-robin assignment of work items (pair indices) to ranks: i -> i % world."""
    return list(range(rank, n_items, world))


def run_pairs(pair_ids, submit, wait, inflight=1):
    """Pushes the pairs through `inflight` pipelines round-robin: pipeline s = i % inflight takes pair i as soon as its
    previous pair has been collected.  submit(slot, pair_id) enqueues (asynchronously if the matcher can),
    wait(slot) completes the pair that is in flight on that slot.  Returns the number of pairs completed."""
    inflight = max(1, int(inflight))
    busy = [False] * inflight
    done = 0
    for i, pid in enumerate(pair_ids):
        s = i % inflight
        if busy[s]:
            wait(s)
            done += 1
        submit(s, pid)
        busy[s] = True
    for s in range(inflight):  # collect in submission order
        k = (len(pair_ids) + s) % inflight
        if busy[k]:
            wait(k)
            busy[k] = False
            done += 1
    return done


class MatchFailed(RuntimeError):
    """Raised by a matcher's submit()/wait() when a Match failed on this GPU (HIP error, lost device, ...)."""


class LocalStore:
    """In-process stand-in for a torch.distributed Store (one rank, or threads in tests): add / set / get / compare_set."""

    def __init__(self):
        import threading
        self._d, self._mu = {}, threading.Lock()

    def add(self, key, amount):
        with self._mu:
            v = int(self._d.get(key, b"0")) + int(amount)
            self._d[key] = str(v).encode()
            return v

    def set(self, key, value):
        with self._mu:
            self._d[key] = value if isinstance(value, bytes) else str(value).encode()

    def get(self, key):
        with self._mu:
            return self._d[key]

    def compare_set(self, key, expected, desired):
        with self._mu:
            cur = self._d.get(key)
            exp = expected if isinstance(expected, bytes) else str(expected).encode()
            if (cur is None and exp == b"") or cur == exp:
                self._d[key] = desired if isinstance(desired, bytes) else str(desired).encode()
            return self._d.get(key, b"")


class PullQueue:
    """Work queue of pair indices 0..n-1 on a Store (torch.distributed TCPStore / PrefixStore, or LocalStore).

    keys: next (pull counter), done (completed pairs), rq_tail / rq_head / rq/<k> (re-queued pairs), failed (retired ranks).
    try_pull() never blocks: it returns a pair index, or None when there is nothing to hand out right now."""

    def __init__(self, store, n_items, world=1):
        self.store, self.n, self.world = store, int(n_items), int(world)
        self._main_exhausted = self.n == 0
        # (idempotent initialisation of the CAS-managed head: every rank may do it)
        self.store.compare_set("rq_head", "", "0")
        # A claim writes "<index>|<claimant>" with a token no other claimant uses: compare_set returns the CURRENT value on
        # failure too, so a bare "<index>" would look like success to a rank that lost the race to exactly one other rank
        # (both would then compute the same re-queued pair and `done` would be over-counted).
        import uuid
        self._token = uuid.uuid4().hex[:16]

    @staticmethod
    def _head_index(value):
        v = value.decode() if isinstance(value, bytes) else str(value)
        return int(v.split("|", 1)[0])

    def try_pull(self):
        if not self._main_exhausted:
            v = self.store.add("next", 1) - 1
            if v < self.n:
                return v
            self._main_exhausted = True
        while True:  # re-queued pairs: claim index head+1 with a compare-and-set
            tail = self.store.add("rq_tail", 0)
            cur = self.store.get("rq_head")
            cur = cur.decode() if isinstance(cur, bytes) else str(cur)
            head = self._head_index(cur)
            if head >= tail:
                return None
            mine = "%d|%s" % (head + 1, self._token)
            got = self.store.compare_set("rq_head", cur, mine)
            got = got.decode() if isinstance(got, bytes) else str(got)
            if got == mine:  # (the whole value, token included: unambiguous)
                key = "rq/%d" % (head + 1)
                for _ in range(20000):  # the writer bumps rq_tail before it sets the entry: wait for the entry
                    try:
                        if hasattr(self.store, "check") and not self.store.check([key]):
                            raise KeyError(key)
                        return int(self.store.get(key))
                    except KeyError:
                        time.sleep(0.0005)
                raise RuntimeError("PullQueue: re-queued entry %s never appeared" % key)

    def requeue(self, pid):
        k = self.store.add("rq_tail", 1)
        self.store.set("rq/%d" % k, str(int(pid)))

    def mark_done(self, n=1):
        return self.store.add("done", int(n))

    def retire(self):
        """This rank stops pulling (its GPU failed).  Returns the number of retired ranks."""
        return self.store.add("failed", 1)

    def all_done(self):
        return self.store.add("done", 0) >= self.n

    def all_failed(self):
        return self.store.add("failed", 0) >= self.world


def run_queue(queue, submit, wait, inflight=1, poll_s=0.001, on_done=None):
    """Pulls pairs from `queue` through `inflight` pipelines until the whole batch is done (by all ranks together).
    submit(slot, pid) / wait(slot) as in run_pairs; either may raise MatchFailed: the pairs in flight on this rank go back
    into the queue, the rank retires and returns.  Returns (pair ids completed here in completion order, failed: bool)."""
    inflight = max(1, int(inflight))
    slots = [None] * inflight
    order = []  # slots in submission order
    mine = []

    def give_up(extra=None):
        for s in order:
            queue.requeue(slots[s])
            slots[s] = None
        del order[:]
        if extra is not None:
            queue.requeue(extra)
        queue.retire()

    def collect_oldest():
        s = order.pop(0)
        pid = slots[s]
        try:
            wait(s)
        except MatchFailed:
            slots[s] = None
            give_up(pid)
            return False
        slots[s] = None
        mine.append(pid)
        queue.mark_done(1)
        if on_done is not None:
            on_done(s, pid)
        return True

    while True:
        free = [s for s in range(inflight) if slots[s] is None]
        if not free:
            if not collect_oldest():
                return mine, True
            continue
        pid = queue.try_pull()
        if pid is None:
            if order:
                if not collect_oldest():
                    return mine, True
                continue
            if queue.all_done():
                return mine, False
            if queue.all_failed():
                raise RuntimeError("pair farm: every rank has retired, %d pairs unfinished" % (queue.n - queue.store.add("done", 0)))
            time.sleep(poll_s)  # another rank may still fail and put its pairs back
            continue
        s = free[0]
        try:
            submit(s, pid)
        except MatchFailed:
            give_up(pid)
            return mine, True
        slots[s] = pid
        order.append(s)


def timed_region(run_steps, steps, warmup, dist=None, device_sync=None, tensor_device="cpu"):
    """Runs `run_steps(n)` for warmup then for `steps` steps bracketed by barrier + device sync on both
    sides; returns (elapsed seconds MAXed over ranks, total steps SUMmed over ranks)."""
    def sync():
        if dist is not None:
            dist.barrier()
        if device_sync is not None:
            device_sync()

    if warmup > 0:
        run_steps(warmup)
    sync()
    t0 = time.perf_counter()
    run_steps(steps)
    if device_sync is not None:
        device_sync()
    if dist is not None:
        dist.barrier()
    if device_sync is not None:
        device_sync()
    elapsed = time.perf_counter() - t0
    total = steps
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=tensor_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([steps], dtype=torch.int64, device=tensor_device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total = int(c.item())
    return elapsed, total


def digest(array_bytes):
    return hashlib.sha256(array_bytes).hexdigest()


def done_counter(n_local, dist=None, tensor_device="cpu"):
    """The completion barrier of the farm: SUM over ranks of the pairs each rank finished."""
    if dist is None:
        return int(n_local)
    import torch
    c = torch.tensor([int(n_local)], dtype=torch.int64, device=tensor_device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return int(c.item())


def gather_digests(local, dist=None):
    """local: {pair id: digest}.  Returns the list of every rank's dict (index = rank)."""
    if dist is None:
        return [dict(local)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, dict(local))
    return out


def cross_check(primary, recheck, reference=None, selfcheck=None):
    """primary / recheck: lists (one dict per rank) of {pair id: digest}: `primary` = the timed batch (every pair exactly
    once over all ranks), `recheck` = each rank's untimed recomputation of ANOTHER rank's pairs.
    reference: optional {pair id: digest} table of the REFERENCE CPU program's outputs for these pairs
    (tests/golden/farm_ref_digests.json, made by tools/make_farm_ref_digests.py from the reference build: the parity claim of the
    batch).  selfcheck: optional table of 1-GPU outputs of THIS product committed earlier
    (tests/golden/farm_selfcheck_digests.json: repeatability across boxes and rounds for pair ids the reference table does not
    cover -- not a parity claim).  Returns a report dict; report["mismatches"] lists offending pair ids."""
    first = {}
    dup = []
    for d in primary:
        for k, v in d.items():
            if k in first:
                dup.append(k)
            first[k] = v
    mism, checked = [], 0
    for d in recheck:
        for k, v in d.items():
            if k in first:
                checked += 1
                if first[k] != v:
                    mism.append(k)

    def against(table):
        n, bad = 0, []
        if table:
            for k, v in first.items():
                r = table.get(str(k), table.get(k))
                if r is not None:
                    n += 1
                    if r != v:
                        bad.append(k)
        return n, sorted(bad)
    ref_checked, ref_mism = against(reference)
    self_checked, self_mism = against(selfcheck)
    return {"pairs": len(first), "duplicates": sorted(dup), "cross_checked": checked, "mismatches": sorted(set(mism)),
            "reference_checked": ref_checked, "reference_mismatches": ref_mism,
            "selfcheck_1gpu_checked": self_checked, "selfcheck_1gpu_mismatches": self_mism}


def load_digest_tables(root, workload, size):
    """(reference table, self-check table) for a workload at `size` = [W, H, D]; None where there is none."""
    import json
    import os
    ref = selfc = None
    try:
        with open(os.path.join(root, "tests", "golden", "farm_ref_digests.json")) as f:
            t = json.load(f)
        if t.get("size") == list(size):
            ref = t.get(workload) or None
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(root, "tests", "golden", "farm_selfcheck_digests.json")) as f:
            t = json.load(f)
        if t.get("workload") == workload and t.get("size") == list(size):
            selfc = t["digests"]
    except (OSError, ValueError):
        pass
    return ref, selfc


def neighbour_pairs(n_items, world, rank):
    """The pairs rank `rank` recomputes after the timed region: those of the next rank (every pair is then computed on
    two different GPUs; with one rank it recomputes its own, i.e. a repeatability check)."""
    return partition(n_items, world, (rank + 1) % world)


def recheck_assignment(all_primary, rank):
    """Pull-queue runs: which pairs rank `rank` recomputes after the timed region = the pairs the NEXT rank (cyclically, among
    the ranks that computed anything) delivered, so every output is computed on two different GPUs when there are two."""
    owners = [r for r, d in enumerate(all_primary) if d]
    if not owners:
        return []
    if rank not in owners:
        return []
    nxt = owners[(owners.index(rank) + 1) % len(owners)]
    return sorted(all_primary[nxt])


def job_store(dist, name):
    """The store the pull queue lives on: the process group's own rendezvous store (TCP, hosted by rank 0) under a prefix."""
    import torch.distributed as td
    base = td.distributed_c10d._get_default_store()
    return td.PrefixStore("adc_farm/%s" % name, base)
