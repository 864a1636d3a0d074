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


def cross_check(primary, recheck, reference=None):
    """primary / recheck: lists (one dict per rank) of {pair id: digest}: `primary` = the timed batch (every pair exactly
    once over all ranks), `recheck` = each rank's untimed recomputation of ANOTHER rank's pairs.  reference: optional
    {pair id: digest} table of 1-GPU outputs.  Returns a report dict; report["mismatches"] lists offending pair ids."""
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
    ref_checked, ref_mism = 0, []
    if reference:
        for k, v in first.items():
            r = reference.get(str(k), reference.get(k))
            if r is not None:
                ref_checked += 1
                if r != v:
                    ref_mism.append(k)
    return {"pairs": len(first), "duplicates": sorted(dup), "cross_checked": checked, "mismatches": sorted(set(mism)),
            "reference_checked": ref_checked, "reference_mismatches": sorted(ref_mism)}


def neighbour_pairs(n_items, world, rank):
    """The pairs rank `rank` recomputes after the timed region: those of the next rank (every pair is then computed on
    two different GPUs; with one rank it recomputes its own, i.e. a repeatability check)."""
    return partition(n_items, world, (rank + 1) % world)
