"""Pair farm: independent stereo pairs are the unit of multi-GPU work.

One process per GPU (torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests).
There is NO data-path collective: every Match is independent.  The only communication is
  * a completion barrier before / after the timed region, and
  * a MAX all-reduce of the per-rank elapsed time (the job is as slow as its slowest GPU),
plus an optional SUM all-reduce of the processed-pair counter (the "trivial completion barrier").
This module holds exactly that logic so that bench.py (HIP runner, RCCL) and the CPU test tier
(fake runner, gloo, world_size 2) exercise the same code.
"""
import time


def partition(n_items, world, rank):
    """Static round-robin assignment of work items (pair indices) to ranks: i -> i % world."""
    return list(range(rank, n_items, world))


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
