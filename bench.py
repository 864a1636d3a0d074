#!/usr/bin/env python3
"""bench.py -- stereo pairs/s of the MI355X-native AD-Census Match path (BASELINE.json metric).

A "step" = one full `Match` (cost volume -> 4x cross aggregation -> 4 scanline passes -> L/R WTA ->
multi-step refinement) of one 1920x1080, D=128 stereo pair whose images are already resident in
HBM (adc_match_device); the disparity map stays in HBM.  N>1: one process per GPU
(torch.distributed / RCCL), pairs are independent work items (weak scaling, no data collective;
RCCL is used only for the barrier and the max-over-ranks reduction of the elapsed time).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the aggregation pass kernel (k_agg_march; k_agg_regring on long-arm images): algorithmic bytes per launch
                  (2*V + 4*P arms [+ 2*P counts on dividing passes], V = 4*W*H*D) / its average
                  launch duration measured with HIP events on the handle's own stream inside the
                  timed region, vs 8 TB/s HBM3E (regular passes only: the first pass, which computes the
                  matching cost itself and only writes, is not part of the average); device_copy_GBps =
                  a device-to-device copy of one volume measured after the timed region (practical ceiling).
  cpu_baseline -- the reference CPU path (oracle/_ref, kind "reference"; the plain-C port if absent)
                  timed on this host, 1 thread, on a bounded row-strip sample of the same pair.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="noise", choices=["noise", "structured"],
                    help="noise = BASELINE.json configs[3] (default); structured = SURVEY 8d S2 pair")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--disp", type=int, default=128)
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("ADC_BENCH_INFLIGHT", "1")),
                    help="ADCensusStereo objects (streams) in flight per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=540, help="rows of the CPU-baseline sample strip")
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if world > 1:
        # torch first: its bundled HIP runtime (same SONAME) is then shared by the C-ABI library
        import torch as _torch
        import torch.distributed as _dist
        torch, dist = _torch, _dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ADC_BENCH_BACKEND", "nccl")  # "gloo": test hook (several ranks on one GPU)
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = int(os.environ.get("ADC_BENCH_DEVICE", local_rank))
            dist.init_process_group(backend=backend)
    import adcensus_amd as A
    from adcensus_amd import workloads
    lib = A.lib()
    if A.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible")
    W, H, D = a.width, a.height, a.disp
    P = W * H
    opt = A.ADCensusOption(min_disparity=0, max_disparity=D)

    # synthetic pair(s): distinct seed per rank and per in-flight slot
    F = max(1, a.inflight)
    handles, bufs, pairs = [], [], []
    for i in range(F):
        seed = 12345 + rank * 64 + i
        left, right = (workloads.noise_pair(W, H, seed) if a.workload == "noise"
                       else workloads.structured_pair(W, H, D, seed=777 + rank * 64 + i))
        pairs.append((left, right))
        st = A.ADCensusStereo(device=local_rank)
        if not st.Initialize(W, H, opt):
            raise SystemExit("Initialize failed: " + A.last_error())
        dl, dr, dd = lib.adc_device_malloc(P * 3), lib.adc_device_malloc(P * 3), lib.adc_device_malloc(P * 4)
        assert dl and dr and dd
        assert lib.adc_memcpy_h2d(dl, left.ctypes.data, P * 3) == 0 and lib.adc_memcpy_h2d(dr, right.ctypes.data, P * 3) == 0
        handles.append(st)
        bufs.append((dl, dr, dd))
    handles[0].set_profiling(True)

    def run_steps(nsteps, collect=None):
        """nsteps Match calls spread over the F objects, one host thread per object."""
        counts = [nsteps // F + (1 if i < nsteps % F else 0) for i in range(F)]
        errs = []

        def worker(i):
            st, (dl, dr, dd) = handles[i], bufs[i]
            for _ in range(counts[i]):
                if not (st.match_device(dl, dr, dd) and st.wait()):
                    errs.append(A.last_error())
                    return
                if collect is not None and i == 0:
                    collect.append((st.stage_ms(), st.aggregate_pass_ms()))
        ths = [threading.Thread(target=worker, args=(i,)) for i in range(F) if counts[i]]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise SystemExit("Match failed: %s" % errs[0])

    from adcensus_amd import farm
    prof = []
    elapsed, total_steps = farm.timed_region(lambda n: run_steps(n, prof), a.steps, a.warmup, dist=dist,
                                             device_sync=lib.adc_device_synchronize,
                                             tensor_device="cuda" if (dist is not None and dist.get_backend() == "nccl") else "cpu")
    prof = prof[-a.steps:] if len(prof) > a.steps else prof  # drop the warm-up samples of object 0

    if rank == 0:
        total_pairs = total_steps  # SUM over ranks of the steps each rank timed
        value = total_pairs / elapsed
        # roofline of the dominant kernel (aggregation pass): algorithmic bytes per launch / avg launch time
        V = 4.0 * P * D
        # a regular pass moves 2V + 4P (+2P counts when it divides); 8 passes = 16V + 32P + 8P.  With the
        # cost fused into the first pass (default) that pass is write-only and is left out of the average: the 7
        # regular passes (4 of them dividing) move 14V + 28P + 8P.
        # Short-arm images (the noise pair): the dividing pass of an iteration and the first pass of the next one share
        # a launch, so after the fused first pass there are 4 launches (3 pairs + the last pass), each read V + write V
        # + arm records and counts = 2V + 6P.
        npass = max([p[1][1] for p in prof] or [8])
        if npass == 7:
            per_launch_bytes = (14.0 * V + 36.0 * P) / 7.0
        elif npass in (4, 5):
            per_launch_bytes = 2.0 * V + 6.0 * P
        else:
            per_launch_bytes = (16.0 * V + 40.0 * P) / 8.0
        agg = [p[1][0] for p in prof if p[1][1] > 0 and p[1][0] > 0]
        agg_ms = float(np.mean(agg)) if agg else float("nan")
        achieved = per_launch_bytes / (agg_ms * 1e-3) / 1e9 if agg else float("nan")
        stage = {}
        if prof:
            for k in prof[0][0]:
                stage[k] = round(float(np.mean([p[0][k] for p in prof])), 4)
        # practical ceiling next to the 8 TB/s peak: a plain device-to-device copy of one volume (read V + write V),
        # measured after the timed region on two scratch buffers
        copy_gbps = None
        try:
            nb = int(V)
            ca, cb = lib.adc_device_malloc(nb), lib.adc_device_malloc(nb)
            if ca and cb:
                ms = lib.adc_device_copy_ms(cb, ca, nb, 5)
                if ms > 0:
                    copy_gbps = round(2.0 * nb / (ms * 1e-3) / 1e9, 1)
            for q in (ca, cb):
                if q:
                    lib.adc_device_free(q)
        except Exception:
            copy_gbps = None
        out = {
            "metric": "stereo pairs/s at 1920x1080 D=128 (ADCensusStereo::Match)" if (W, H, D) == (1920, 1080, 128)
                      else "stereo pairs/s at %dx%d D=%d (ADCensusStereo::Match)" % (W, H, D),
            "value": round(value, 4), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1000.0 * elapsed / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d D=%d%s" % (a.workload, W, H, D, {(1920, 1080, 128): " (BASELINE.json configs[3])" if a.workload == "noise" else " (size of BASELINE.json configs[3], SURVEY 8d structured pair)",
                                                                            (1242, 375, 128): " (size of BASELINE.json configs[2])",
                                                                            (450, 375, 64): " (size of BASELINE.json configs[1])"}.get((W, H, D), "")),
                       "in_flight_per_gpu": F, "parallelism": "replicas x%d (independent pairs)" % world},
            "ms_per_pair_latency": round(float(np.sum(list(stage.values()))), 4) if stage else None,
            "stage_ms": stage,
            "roofline": {"kernel": "%s (one aggregation launch: %s)" % (("k_agg_march", "pass pair, 2 passes of work") if npass in (4, 5)
                                                                                    else ("k_agg_regring / k_agg_march", "one pass")), "bound": "hbm",
                         "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 4), "traffic": pmc_traffic(a.workload, (W, H, D)),
                         "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": round(agg_ms, 5),
                         "device_copy_GBps": copy_gbps,
                         "frac_of_device_copy": round(achieved / copy_gbps, 4) if copy_gbps else None},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pairs[0], D, a.cpu_rows, H)
        print(json.dumps(out), flush=True)

    for st, (dl, dr, dd) in zip(handles, bufs):
        st.Release()
        for p in (dl, dr, dd):
            lib.adc_device_free(p)
    if dist is not None:
        dist.destroy_process_group()


def pmc_traffic(workload, whd):
    """HBM bytes per aggregation launch from the committed rocprofv3 PMC passes (profiles/, FETCH_SIZE x2 +
    WRITE_SIZE, separate --pmc runs, gfx950 correction per MI355X_MICROARCH.md); None when not collected for
    this workload / size."""
    if whd != (1920, 1080, 128):
        return None
    p = os.path.join(ROOT, "profiles", "r1_k4_pmc_traffic_%s.json" % workload)
    try:
        with open(p) as f:
            return float(json.load(f)["traffic_bytes_per_launch_avg"])
    except Exception:
        return None


def cpu_baseline(pair, D, rows, H):
    """Reference CPU path on a bounded sample: the top `rows` rows of the same pair, full width and
    disparity range, 1 thread (the reference is single-threaded).  pairs/s is scaled by rows/H."""
    from oracle import pyoracle  # checker / baseline leg only
    orc = pyoracle.load("auto")
    rows = min(rows, H)
    l, r = np.ascontiguousarray(pair[0][:rows]), np.ascontiguousarray(pair[1][:rows])
    opt = pyoracle.Option(max_disparity=D)
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    sys.stdout.flush()
    os.dup2(devnull, 1)  # the reference printf()s its stage timings
    try:
        _, secs = orc.match(l, r, opt)
    finally:
        sys.stdout.flush()
        try:  # the reference printf()s through the C library's own buffer: flush it while fd 1 is still /dev/null,
            import ctypes  # otherwise those lines appear after the JSON line at process exit
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)
    full_secs = secs * (H / float(rows))
    return {"value": round(1.0 / full_secs, 6), "unit": "pairs/s", "cores": 1, "kind": "reference" if orc.kind == "reference" else "port",
            "sample": "top %d of %d rows of the same pair (full width, D=%d): %.2f s measured, scaled by rows to %.1f s/pair; "
                      "host has %d cores, the reference is single-threaded" % (rows, H, D, secs, full_secs, os.cpu_count() or 0)}


if __name__ == "__main__":
    main()
