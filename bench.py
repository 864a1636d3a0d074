#!/usr/bin/env python3
"""bench.py -- stereo pairs/s of the MI355X-native AD-Census Match path (BASELINE.json metric).

A "step" = one full `Match` (cost volume -> 4x cross aggregation -> 4 scanline passes -> L/R WTA -> multi-step
refinement) of one 1920x1080, D=128 stereo pair whose images are already resident in HBM (adc_match_device); the
disparity map stays in HBM.  The timed region is a FARM over a batch of DISTINCT pairs (BASELINE.json configs[4],
SURVEY.md 8d config 5), pair i = the seeded synthetic pair 12345 + i:
  default       weak scaling: batch = steps x ranks pairs, static partition (rank r takes the pairs i = r mod ranks),
                every GPU does exactly `steps` pairs;
  --batch B     strong scaling = configs[4] literally: a FIXED batch of B pairs (64), every rank holds all B pairs in its
                HBM and PULLS the next pair index from one shared counter on the job's rendezvous store
                (adcensus_amd/farm.py PullQueue; a failed Match is re-queued and the rank retires).  steps := B.
`--inflight` pipelines (ADCensusStereo objects on separate streams) per GPU: default 1 for one GPU (clean per-kernel
timings for the roofline), 2 for several.  There is no data-path collective: RCCL carries the barriers, the MAX over ranks
of the elapsed time, the SUM of the done counter and an all-gather of one SHA-256 per pair.  After the timed region every
output of the batch is recomputed on another GPU (N = 1: a second time) and the digests compared; they are also compared
with the digests of the REFERENCE CPU program's maps for the same pairs (tests/golden/farm_ref_digests.json) when the size matches.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one rank per GPU).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline       the aggregation kernel (K4): achieved = the bytes a regular launch really moves through HBM (read V +
                 write V + records; a pass-pair launch does that once for two passes of work) / the average launch
                 duration from HIP events on the handle's own stream inside the timed region; frac = achieved / 8 TB/s
                 (<= 1 by construction).  algorithmic_achieved / algorithmic_frac = the SURVEY.md 8d numerator (every
                 algorithmic pass counted with 2V + 4P [+ 2P]: exceeds the real traffic when launches fuse passes).
                 traffic = PMC bytes per launch (profiles/r<N>_k4_pmc_traffic_<workload>.json, newest round first; None unless that file was
                 measured on exactly these kernel sources -- SHA-256 of k_aggregate*.{hip,h}).  device_copy_GBps = a
                 device-to-device copy of one volume measured after the timed region (the practical ceiling).
  stage_roofline HBM fractions of the scanline stage (4 x (2V + 3P)), the right-view WTA (V) and the whole Match
                 (26 V = 27.6 GB) from the stage timings / ms_per_step.
  structured     (N = 1) the same measurement on the SURVEY 8d "structured" pair (natural-image-like arms / voting load).
  host_inclusive (N = 1) the drop-in entry point adc_match(left, right, disp) with pageable host buffers.
  host_farm      the persistent farm of the C ABI (adc_farm_*) fed from pageable host buffers (N > 1: every rank feeds its
                 share of the batch that way in a second timed region -- what configs[4] describes; PCIe inclusive, never `value`).
  throughput_mode (N = 1) the same batch with 3 pipelines in flight per GPU.
  scaling_reference (N > 1) rank 0 ALONE on its GPU with the same pairs per GPU and the same number of pipelines, measured
                 after the farm (the other ranks idle at a barrier): the 1-GPU number this box gives for the N-GPU line.
  match_host     (N = 1) THE DROP-IN FIGURE: pairs/s and ms/pair of ADCensusStereo::Match == adc_match(host, host, host), pageable
                 and caller-registered buffers, both workloads (`value` above is adc_match_device: images / map resident in HBM).
  cpu_baseline   the reference CPU path (oracle/_ref, kind "reference"; the plain-C port if absent) timed on this host,
                 1 thread, on the WHOLE pair 0 of the batch (--cpu-rows R: the top R rows, scaled).
  cpu_baseline_all_cores  N independent reference processes side by side (pairs are independent; SURVEY.md 8d "all host cores"),
                 N stated; N = min(16, host cores, what the free memory holds at 2.6 GB each).
Exit code: 0; 3 when RCCL was requested and the job had to run its barrier / reductions over gloo instead (the line is still
printed, with config.comm_backend saying so) -- a broken RCCL must not look like a pass.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

K4_SOURCES = ("k_aggregate.hip", "k_aggregate_rr.h", "k_aggregate_rr2.h")


def k4_source_hash():
    """SHA-256 (first 16 hex digits) over the aggregation kernel sources: a committed PMC measurement is only reported
    as `roofline.traffic` when it was taken on exactly these sources (tools/pmc_summary.py stores the same hash)."""
    import hashlib
    h = hashlib.sha256()
    for name in K4_SOURCES:
        with open(os.path.join(ROOT, "adcensus_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


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
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("ADC_BENCH_INFLIGHT", "0")),
                    help="ADCensusStereo objects (streams) in flight per GPU in the timed region (0 = 1 for one GPU, 2 for several)")
    ap.add_argument("--batch", type=int, default=0,
                    help="fixed batch of B distinct pairs farmed over all GPUs through the pull queue (strong scaling, "
                         "BASELINE.json configs[4]: --batch 64); 0 = steps x ranks pairs, static partition (weak scaling)")
    ap.add_argument("--queue", default="", choices=["", "static", "pull"], help="how the batch is handed out (default: pull with --batch, else static)")
    ap.add_argument("--spinup-ms", type=float, default=float(os.environ.get("ADC_BENCH_SPINUP_MS", "400")),
                    help="untimed device spin-up before the warm-up steps of the headline region: matches are run until this many ms have "
                         "passed, so that the timed region starts at the clocks of a busy GPU (the first 20 ms of work after idle ran "
                         "5 %% slower than the same work a second later)")
    ap.add_argument("--no-host-leg", action="store_true", help="N > 1: skip the second timed region fed from host buffers (adc_farm_*)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the structured / host-inclusive / throughput legs")
    ap.add_argument("--no-mixed-leg", action="store_true", help="skip the alternating structured / noise stream leg")
    ap.add_argument("--check-pairs", type=int, default=10,
                    help="distinct pairs of the second-workload leg, each checked against the reference digest table (structured pairs take seconds to generate)")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU-baseline sample strip (0 = the whole frame, ~20 s)")
    ap.add_argument("--cpu-baseline-structured", action="store_true",
                    help="also time the reference on the structured pair (SURVEY 8d S2; ~35 s on one host core) -> structured.cpu_baseline")
    ap.add_argument("--cpu-all-cores", type=int, default=16,
                    help="processes of the all-host-cores CPU figure (0 = skip; clamped to the host's cores and free memory)")
    ap.add_argument("--no-cone-leg", action="store_true", help="skip the Cone 450x375 D=64 leg (BASELINE.json configs[0] / [1])")
    ap.add_argument("--write-digests", default="", help="write {pair id: sha256} of the batch outputs to this file (N = 1)")
    return ap.parse_args()


def reexec_under_torchrun(a):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


class Matcher:
    """`inflight` ADCensusStereo pipelines on one GPU + the device-resident inputs / outputs of a list of pairs."""

    def __init__(self, A, device, W, H, D, inflight):
        self.A, self.lib = A, A.lib()
        self.W, self.H, self.D, self.P = W, H, D, W * H
        self.opt = A.ADCensusOption(min_disparity=0, max_disparity=D)
        self.handles = []
        for _ in range(max(1, inflight)):
            st = A.ADCensusStereo(device=device)
            if not st.Initialize(W, H, self.opt):
                raise SystemExit("Initialize failed: " + A.last_error())
            self.handles.append(st)
        self.buf = {}  # pair id -> (d_left, d_right, d_disp)
        self.soft_fail = False  # True: a failed Match raises farm.MatchFailed (pull queue: re-queue + retire) instead of aborting

    def _fail(self, what):
        if self.soft_fail:
            from adcensus_amd import farm
            raise farm.MatchFailed(what + ": " + self.A.last_error())
        raise SystemExit(what + ": " + self.A.last_error())

    def upload(self, pid, left, right):
        lib, P = self.lib, self.P
        dl, dr, dd = lib.adc_device_malloc(P * 3), lib.adc_device_malloc(P * 3), lib.adc_device_malloc(P * 4)
        assert dl and dr and dd, "device allocation failed"
        assert lib.adc_memcpy_h2d(dl, left.ctypes.data, P * 3) == 0 and lib.adc_memcpy_h2d(dr, right.ctypes.data, P * 3) == 0
        self.buf[pid] = (dl, dr, dd)

    def submit(self, slot, pid):
        dl, dr, dd = self.buf[pid]
        if not self.handles[slot].match_device(dl, dr, dd):
            self._fail("Match failed")

    def wait(self, slot):
        if not self.handles[slot].wait():
            self._fail("Match failed")

    def output(self, pid):
        out = np.empty((self.H, self.W), np.float32)
        assert self.lib.adc_memcpy_d2h(out.ctypes.data, self.buf[pid][2], self.P * 4) == 0
        return out

    def clear_output(self, pid):
        z = np.full((self.H, self.W), -1.0, np.float32)
        assert self.lib.adc_memcpy_h2d(self.buf[pid][2], z.ctypes.data, self.P * 4) == 0

    def free(self, pid):
        for p in self.buf.pop(pid):
            self.lib.adc_device_free(p)

    def release(self):
        for pid in list(self.buf):
            self.free(pid)
        for st in self.handles:
            st.Release()


class HostFedMatcher:
    """The persistent farm of the C ABI (adc_farm_*: pinned staging ring, asynchronous submit, ordered delivery) fed with
    pageable host images, delivering into pageable host maps -- the N > 1 runner of BASELINE.json configs[4]."""

    def __init__(self, A, device, W, H, D, inflight, pairs):
        self.A, self.W, self.H = A, W, H
        self.farm = A.PairFarm(W, H, A.ADCensusOption(min_disparity=0, max_disparity=D), device=device, pipelines=max(1, inflight))
        self.pairs = pairs  # pid -> (left, right) host arrays
        self.out = {pid: np.empty((H, W), np.float32) for pid in pairs}
        self.ticket = {}

    def submit(self, slot, pid):
        from adcensus_amd import farm
        l, r = self.pairs[pid]
        try:
            self.ticket[slot] = self.farm.submit(l, r, self.out[pid])
        except self.A.PreviousPairFailed as e:
            # the NEW pair is in flight (its ticket is valid: a later wait(slot) finds it); the pipeline's previous pair failed:
            # the pull queue re-queues what this rank has in flight and retires the rank
            self.ticket[slot] = e.ticket
            raise farm.MatchFailed(str(e))
        except RuntimeError as e:
            raise farm.MatchFailed(str(e))

    def wait(self, slot):
        from adcensus_amd import farm
        try:
            self.farm.wait(self.ticket.pop(slot))
        except RuntimeError as e:
            raise farm.MatchFailed(str(e))

    def release(self):
        self.farm.close()


def make_pair(workload, W, H, D, pid):
    from adcensus_amd import workloads
    if workload == "noise":
        return workloads.noise_pair(W, H, 12345 + pid)  # seeds 12345 + i
    return workloads.structured_pair(W, H, D, seed=777 + pid)


def device_copy_rate(lib, nbytes):
    """Device-to-device copy of one volume (read nbytes + write nbytes), GB/s, or None: (best, hipMemcpyAsync, copy kernel).
    The kernel is the hardware guide's yardstick shape (float4 grid-stride copy, best of a few grids, plain / non-temporal)."""
    try:
        ca, cb = lib.adc_device_malloc(nbytes), lib.adc_device_malloc(nbytes)
        rates = [None, None]
        if ca and cb:
            for k, name in enumerate(("adc_device_copy_ms", "adc_device_copy_kernel_ms")):
                if not hasattr(lib, name):
                    continue
                t = getattr(lib, name)(cb, ca, nbytes, 5)
                if t > 0:
                    rates[k] = round(2.0 * nbytes / (t * 1e-3) / 1e9, 1)
        for q in (ca, cb):
            if q:
                lib.adc_device_free(q)
        have = [r for r in rates if r]
        return (max(have) if have else None, rates[0], rates[1])
    except Exception:
        return (None, None, None)


def k4_roofline(prof, W, H, D, lib, workload, kernel=None, in_flight=1):
    """prof: list of aggregate_info() tuples of the profiled handle, one per timed Match."""
    P = float(W) * H
    V = 4.0 * P * D
    prof = [p for p in prof if p[1] > 0 and p[0] > 0]
    if not prof:
        return None
    ms = float(np.mean([p[0] for p in prof]))
    launches, passes, fused = prof[-1][1], prof[-1][2], prof[-1][3]
    # algorithmic bytes of the regular launches: every pass 2V + 4P; the dividing ones (4 of 8, all regular) + 2P
    alg_total = passes * (2.0 * V + 4.0 * P) + 4 * 2.0 * P
    hbm_total = launches * (2.0 * V + 4.0 * P) + 4 * 2.0 * P  # a pair launch still reads V and writes V once
    alg_per_launch, hbm_per_launch = alg_total / launches, hbm_total / launches
    alg = alg_per_launch / (ms * 1e-3) / 1e9
    hbm = hbm_per_launch / (ms * 1e-3) / 1e9
    copy_gbps, copy_memcpy, copy_kernel = device_copy_rate(lib, int(V))
    pairs = passes > launches
    if not kernel:
        kernel = "pass-pair launches" if pairs else "one pass per launch"
    elif pairs:  # the library names the family of the LAST regular launch; say that most launches of this Match fuse two passes
        kernel = kernel.replace("one pass per launch", "pass-pair launches: %.2f passes per launch" % (passes / float(launches)))
    traffic = pmc_traffic(workload, (W, H, D))
    return {"kernel": kernel, "bound": "hbm", "achieved": round(hbm, 2), "peak": 8000.0, "unit": "GB/s",
            "frac": round(hbm / 8000.0, 4), "traffic": traffic,
            "bytes_per_launch": hbm_per_launch, "avg_launch_ms": round(ms, 5), "regular_launches": launches,
            "passes_per_launch": round(passes / float(launches), 3), "first_pass_fused_with_cost": bool(fused),
            "algorithmic_bytes_per_launch": alg_per_launch, "algorithmic_achieved": round(alg, 2),
            "algorithmic_frac": round(alg / 8000.0, 4),
            "traffic_over_bytes": round(traffic / hbm_per_launch, 4) if traffic else None,
            "device_copy_GBps": copy_gbps, "frac_of_device_copy": round(hbm / copy_gbps, 4) if copy_gbps else None,
            "device_copy_detail_GBps": {"hipMemcpyAsync": copy_memcpy, "float4_grid_stride_kernel": copy_kernel},
            "in_flight_while_measured": in_flight,
            "note": "achieved = HBM bytes a regular launch moves (read V + write V + records) / HIP-event launch-to-launch time; "
                    "algorithmic_* = SURVEY 8d numerator (2V+4P[+2P] per algorithmic pass; a pass-pair launch covers two)"}


def stage_roofline(stage, ms_per_step, W, H, D, fused_tail=False):
    """HBM fractions of the other volume stages from the stage timings (events on the handle's stream) and of the whole Match.
    fused_tail: the last aggregation pass ran inside the first scanline pass (short-arm images, k_scanline_seg_agg): the
    aggregation stage then covers 7 of the 8 algorithmic passes; the scanline stage's traffic is unchanged."""
    P = float(W) * H
    V = 4.0 * P * D
    out = {}

    def frac(nbytes, ms):
        return {"bytes": nbytes, "ms": round(ms, 4), "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac": round(nbytes / (ms * 1e-3) / 8e12, 4)}
    if stage.get("scanline", 0) > 0:
        out["scanline_K5"] = dict(frac(4 * (2.0 * V + 3.0 * P), stage["scanline"]), what="4 chained DP passes, each reads V + writes V + 3P of penalty inputs")
    if stage.get("wta", 0) > 0:
        out["wta_right_K6"] = dict(frac(V, stage["wta"]), what="right-view winner-takes-all: one read of V (the left view rides on the last scanline pass)")
    if stage.get("aggregate", 0) > 0:
        if fused_tail:
            out["aggregate_K4_stage_algorithmic"] = dict(frac(14.0 * V + 34.0 * P, stage["aggregate"]),
                                                         what="7 of the 8 passes x (2V + 4P) + 6P (the 8th runs inside the first scanline pass) / stage time")
        else:
            out["aggregate_K4_stage_algorithmic"] = dict(frac(16.0 * V + 40.0 * P, stage["aggregate"]),
                                                         what="SURVEY 8d stage figure: 8 passes x (2V + 4P) + 8P = 17.07 GB at 1080p / stage time (target >= 0.60)")
    if ms_per_step and ms_per_step > 0:
        out["whole_match"] = dict(frac(26.0 * V, ms_per_step), what="SURVEY 8d whole-pipeline model 26 V (27.6 GB at 1080p/128) / ms_per_step")
    return out


def measure_workload(A, device, W, H, D, workload, steps, warmup, inflight, pair_ids, dist=None, tensor_device="cpu",
                     queue_factory=None, host_pairs=None, spinup_ms=0.0):
    """Uploads the pairs, runs warm-up + the timed farm region; returns (matcher, elapsed, total, stage_ms, roofline-prof).
    queue_factory: None = static list `pair_ids` repeated to `steps`; else a callable returning a farm.PullQueue for the
    timed region (the batch = range(queue.n): every pair must be in `pair_ids`)."""
    from adcensus_amd import farm
    if host_pairs is not None:
        m = HostFedMatcher(A, device, W, H, D, inflight, host_pairs)
    else:
        m = Matcher(A, device, W, H, D, inflight)
        for pid in pair_ids:
            l, r = make_pair(workload, W, H, D, pid)
            m.upload(pid, l, r)
        # the timed region records only the marks around the aggregation launches (the live duration of the roofline kernel); the
        # stage times come from extra Matches BEHIND the region with the stage marks on (an event record costs the stream ~6 us:
        # ten of them are 1.3 % of a 1080p Match -- instrumentation the product path does not have)
        m.handles[0].set_profiling(2)
    prof, stages = [], []
    m.mine, m.retired = list(pair_ids), False

    def on_collected(slot):
        if slot == 0 and host_pairs is None:
            prof.append(m.handles[0].aggregate_info())
            stages.append(m.handles[0].stage_ms())

    def wait(slot):
        m.wait(slot)
        on_collected(slot)

    def run(n):
        if queue_factory is not None and run.timed:
            m.soft_fail = True
            m.mine, m.retired = farm.run_queue(queue_factory(), m.submit, wait, inflight)
            m.soft_fail = False
            return
        ids = [pair_ids[i % len(pair_ids)] for i in range(n)]
        farm.run_pairs(ids, m.submit, wait, inflight)
    run.timed = False

    def run_timed(n):
        run.timed = True
        run(n)
    if spinup_ms > 0:  # clock ramp (untimed): the same matches, until the time is up
        t_end = time.perf_counter() + spinup_ms * 1e-3
        while time.perf_counter() < t_end:
            run(max(1, inflight))
        del prof[:], stages[:]
    if warmup > 0:
        run(warmup)
    sync = m.lib.adc_device_synchronize if host_pairs is None else A.lib().adc_device_synchronize
    elapsed, total = farm.timed_region(run_timed, steps, 0, dist=dist, device_sync=sync, tensor_device=tensor_device)
    if queue_factory is not None:  # the units really processed: this rank's share of the batch, SUMmed over the ranks
        total = farm.done_counter(len(m.mine), dist, tensor_device)
    keep = max(1, (steps + inflight - 1) // inflight)
    if host_pairs is None:  # stage times: the same pairs once more (at most 10 Matches), untimed, stage marks on
        del stages[:]
        m.handles[0].set_profiling(1)
        run.timed = False
        n_prof = len(prof)
        run(min(max(3, len(pair_ids)), 10) * inflight)
        del prof[n_prof:]  # (the roofline figure stays the timed region's)
        m.handles[0].set_profiling(2)
        sync()
    m.fallbacks = {"median_handoff": 0, "voting_continuations": 0, "aggregation_redos": 0, "scanline_seam_redos": 0,
                   "median_spec_seam_failures": 0, "aggregation_two_plan_matches": 0, "aggregation_tail_in_scanline": 0, "matches": warmup + steps}
    if host_pairs is None:
        # how often adc_wait had to complete an assumption of the asynchronous pipeline (warm-up + timed region, all pipelines)
        for st in m.handles:
            for key, which in (("median_handoff", 0), ("voting_continuations", 1), ("aggregation_redos", 2), ("scanline_seam_redos", 4),
                               ("median_spec_seam_failures", 7), ("aggregation_two_plan_matches", 10), ("aggregation_tail_in_scanline", 13)):
                m.fallbacks[key] += int(st.debug_counter(which))
            m.fallbacks["scanline_segments_per_row"] = int(m.handles[0].debug_counter(5))
            m.fallbacks["voting_chain_budget"] = int(m.handles[0].debug_counter(3))
    return m, elapsed, total, stages[-keep:], prof[-keep:]


def mean_stages(stages):
    return {k: round(float(np.mean([s[k] for s in stages])), 4) for k in stages[0]} if stages else {}


def init_dist(world, local_rank):
    """torch.distributed for world > 1 (or ADC_BENCH_FORCE_DIST=1).  Returns (dist, backend label, tensor device, local_rank,
    what the communicator itself reports)."""
    # torch first: its bundled HIP runtime (same SONAME) is then shared by the C-ABI library
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    backend = os.environ.get("ADC_BENCH_BACKEND", "nccl")  # "gloo": test hook (several ranks on one GPU)
    tensor_device = "cpu"
    seen = None
    if backend == "nccl":
        # RCCL carries a barrier, a few scalar all-reduces and one all-gather of digests: nothing of the data path.  If
        # the communicator cannot be created on this node, the same calls run over gloo instead of failing the job
        try:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            seen = int(probe.item())  # the number of ranks that took part in an RCCL all-reduce
            assert seen == int(os.environ["WORLD_SIZE"])
            tensor_device = "cuda"
        except Exception as exc:  # noqa: BLE001
            print("bench.py: RCCL initialisation failed (%s: %s); falling back to gloo" % (type(exc).__name__, exc), file=sys.stderr)
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
            backend = "gloo (RCCL failed)"
            dist.init_process_group(backend="gloo")
    else:
        local_rank = int(os.environ.get("ADC_BENCH_DEVICE", local_rank))
        dist.init_process_group(backend=backend)
    if seen is None:
        probe = torch.ones(1)
        dist.all_reduce(probe)
        seen = int(probe.item())
    return dist, backend, tensor_device, local_rank, {"backend": backend, "world_size": dist.get_world_size(), "ranks_in_all_reduce": seen}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(reexec_under_torchrun(a))
    # stdout carries exactly ONE line (the JSON of rank 0): everything else that writes to fd 1 underneath us (the RCCL version
    # banner at communicator creation, the reference's printf()s in the CPU-baseline leg) is sent to stderr until then
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist, tensor_device, backend, comm = None, "cpu", None, None
    if world > 1 or os.environ.get("ADC_BENCH_FORCE_DIST") == "1":  # (FORCE_DIST: exercise the RCCL calls with one rank)
        dist, backend, tensor_device, local_rank, comm = init_dist(world, local_rank)
    stub_module = os.environ.get("ADC_BENCH_MATCHER_MODULE", "")
    if stub_module:  # test hook (tests/bench_stub.py): the N > 1 path end to end on a machine without a GPU; never a measurement
        import importlib
        A = importlib.import_module(stub_module)
    else:
        import adcensus_amd as A
    from adcensus_amd import farm
    lib = A.lib()
    if A.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible")
    W, H, D = a.width, a.height, a.disp
    F = a.inflight if a.inflight > 0 else (1 if world == 1 else 2)
    strong = a.batch > 0
    mode = a.queue or ("pull" if strong else "static")

    # ---- the batch
    if strong:
        batch, steps = a.batch, a.batch  # a step = one pair of the fixed batch (whole job)
    else:
        batch, steps = a.steps * world, a.steps
    if mode == "pull":
        mine = list(range(batch))  # every rank holds every pair: any rank may pull any index
        qname = "timed"

        def queue_factory():
            store = farm.job_store(dist, qname) if dist is not None else queue_factory.local
            return farm.PullQueue(store, batch, world=world)
        queue_factory.local = farm.LocalStore()
    else:
        mine = farm.partition(batch, world, rank)
        queue_factory = None
    m, elapsed, total, stages, prof = measure_workload(A, local_rank, W, H, D, a.workload, steps if mode == "static" else batch, a.warmup, F, mine,
                                                       dist=dist, tensor_device=tensor_device, queue_factory=queue_factory, spinup_ms=a.spinup_ms)
    computed = list(m.mine) if mode == "pull" else mine
    # ---- verification (untimed): digests of this rank's outputs, recomputation on another GPU
    primary = {pid: farm.digest(m.output(pid).tobytes()) for pid in computed}
    all_primary = farm.gather_digests(primary, dist)
    todo = farm.recheck_assignment(all_primary, rank) if mode == "pull" else farm.neighbour_pairs(batch, world, rank)
    recheck = {}
    for pid in ([] if m.retired else todo):
        own = pid in m.buf
        if not own:
            l, r = make_pair(a.workload, W, H, D, pid)
            m.upload(pid, l, r)
        else:
            m.clear_output(pid)
        m.submit(0, pid)
        m.wait(0)
        recheck[pid] = farm.digest(m.output(pid).tobytes())
        if not own:
            m.free(pid)
    done = farm.done_counter(len(primary), dist, tensor_device)
    all_recheck = farm.gather_digests(recheck, dist)
    retired = farm.done_counter(1 if m.retired else 0, dist, tensor_device)

    out = None
    if rank == 0:
        ref_table, self_table = farm.load_digest_tables(ROOT, a.workload, [W, H, D])
        check = farm.cross_check(all_primary, all_recheck, ref_table, self_table)
        check["reference"] = "tests/golden/farm_ref_digests.json: SHA-256 of the reference CPU program's maps (oracle/_ref, tools/make_farm_ref_digests.py)"
        check["done_counter"] = done
        check["retired_ranks"] = retired
        check["pairs_per_rank"] = [len(d) for d in all_primary]
        check["ok"] = (done == batch and check["pairs"] == batch and not check["duplicates"] and not check["mismatches"]
                       and not check["reference_mismatches"] and not check["selfcheck_1gpu_mismatches"])
        if a.write_digests and world == 1:
            with open(a.write_digests, "w") as f:
                json.dump({"workload": a.workload, "size": [W, H, D], "generator": "python bench.py --steps %d --write-digests ... (1 GPU)" % a.steps,
                           "digests": {str(k): v for k, v in sorted(primary.items())}}, f, indent=1, sort_keys=True)
        value = total / elapsed
        stage = mean_stages(stages)
        ms_per_step = 1000.0 * elapsed / steps
        sizes = {(1920, 1080, 128): " (BASELINE.json configs[3])" if a.workload == "noise" else " (size of BASELINE.json configs[3], SURVEY 8d structured pair)",
                 (1242, 375, 128): " (size of BASELINE.json configs[2])", (450, 375, 64): " (size of BASELINE.json configs[1])"}
        out = {
            "metric": "stereo pairs/s at %dx%d D=%d (ADCensusStereo::Match, images and disparity map resident in HBM)" % (W, H, D),
            "value": round(value, 4), "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d D=%d%s" % (a.workload, W, H, D, sizes.get((W, H, D), "")),
                       "batch": ("fixed batch of %d distinct pairs (seeds 12345+i) over %d GPU(s) = BASELINE.json configs[4]; a step = one pair of the batch" % (batch, world)) if strong
                                else "%d distinct pairs (seeds 12345+i), %d per GPU" % (batch, a.steps),
                       "queue": "pull (shared counter on the rendezvous store, re-queue on failure)" if mode == "pull" else "static round-robin partition",
                       "in_flight_per_gpu": F, "parallelism": "replicas x%d (independent pairs)" % world,
                       "spinup_ms": a.spinup_ms,
                       "comm_backend": (backend if dist is not None else None), "comm": comm},
            "farm_check": check,
            "device_binding": {"rank": rank, "local_rank_env": int(os.environ.get("LOCAL_RANK", "0")), "device_index": local_rank,
                               "matcher_module": stub_module or "adcensus_amd"},
            "ms_per_pair_latency": round(float(np.sum(list(stage.values()))), 4) if stage else None,
            "stage_ms": stage,
            "stage_ms_source": "Matches behind the timed region with the stage marks on (an event record costs the stream ~6 us; the timed region "
                               "records only the marks around the aggregation launches, which the roofline figure needs)",
            "roofline": k4_roofline(prof, W, H, D, lib, a.workload, m.handles[0].aggregate_kernel(), F),
            # (per-GPU time per Match: weak scaling = ms_per_step, strong = ms_per_step x ranks)
            "stage_roofline": dict(stage_roofline(stage, ms_per_step * (world if strong else 1), W, H, D,
                                                  fused_tail=m.fallbacks.get("aggregation_tail_in_scanline", 0) * 2 > m.fallbacks.get("matches", 1) * len(m.handles)),
                                   **({} if F == 1 else {"note": "%d pipelines in flight: stage times include co-running kernels of other pairs; whole_match is a throughput figure" % F})),
            "async_fallbacks": m.fallbacks,
        }
    host_pairs_all = None
    m.release()

    # ---- N > 1: the same farm fed from pageable host buffers through adc_farm_* (PCIe inclusive; never `value`)
    if world > 1 and not a.no_host_leg:
        ids = list(range(batch)) if mode == "pull" else mine
        host_pairs_all = {pid: make_pair(a.workload, W, H, D, pid) for pid in ids}
        qf2 = None
        if mode == "pull":
            def qf2():
                return farm.PullQueue(farm.job_store(dist, "hostfed"), batch, world=world)
        mh, eh, th, _, _ = measure_workload(A, local_rank, W, H, D, a.workload, steps if mode == "static" else batch, min(a.warmup, 2), F, ids,
                                            dist=dist, tensor_device=tensor_device, queue_factory=qf2, host_pairs=host_pairs_all)
        mh.release()
        if rank == 0:
            out["host_farm"] = {"value": round(th / eh, 4), "unit": "pairs/s", "pipelines_per_gpu": F, "n_gpus": world,
                                "entry_point": "adc_farm_submit / adc_farm_wait on every rank: pageable host images in, pageable host maps out "
                                               "(pinned staging + H2D + kernels + D2H + copy-out inside the timed region)"}
        host_pairs_all = None
    # ---- N > 1: rank 0 alone with the same per-GPU load (the others wait at the barrier)
    if world > 1:
        if rank == 0:
            n1 = steps if not strong else max(1, batch // world)
            m1, e1, t1, _, _ = measure_workload(A, local_rank, W, H, D, a.workload, n1, min(a.warmup, 2), F, list(range(min(n1, 16))))
            out["scaling_reference"] = {"n_gpus": 1, "in_flight_per_gpu": F, "pairs": n1, "value": round(t1 / e1, 4), "unit": "pairs/s",
                                        "note": "rank 0 alone on its GPU, same pipelines and pairs per GPU, measured after the farm in the same process"}
            m1.release()
        dist.barrier()

    if rank == 0 and world == 1 and not a.no_extra_legs:
        other = "structured" if a.workload == "noise" else "noise"
        # ---- second workload (one pair: generating structured pairs costs seconds each), one pipeline
        n2 = max(5, min(10, a.steps))
        ref2, _ = farm.load_digest_tables(ROOT, other, [W, H, D])
        ids2 = list(range(min(n2, len(ref2)))) if (ref2 and a.check_pairs > 0) else [0]
        ids2 = ids2[:max(1, a.check_pairs)]
        m2, e2, t2, st2, pf2 = measure_workload(A, local_rank, W, H, D, other, n2, 2, 1, ids2)
        s2 = mean_stages(st2)
        out[other] = {"value": round(t2 / e2, 4), "unit": "pairs/s", "ms_per_step": round(1000.0 * e2 / n2, 4), "steps": n2,
                      "workload": "%s %dx%d D=%d (%d distinct pairs, seeds %d+i)" % (other, W, H, D, len(ids2), 777 if other == "structured" else 12345),
                      "stage_ms": s2,
                      "roofline": k4_roofline(pf2, W, H, D, lib, other, m2.handles[0].aggregate_kernel(), 1),
                      "stage_roofline": stage_roofline(s2, 1000.0 * e2 / n2, W, H, D,
                                                       fused_tail=m2.fallbacks.get("aggregation_tail_in_scanline", 0) * 2 > m2.fallbacks.get("matches", 1)),
                      "async_fallbacks": m2.fallbacks}
        if ref2:  # every output of the leg against the reference CPU program's map of the same pair
            got2 = {pid: farm.digest(m2.output(pid).tobytes()) for pid in ids2}
            out[other]["reference_check"] = {"pairs": len(got2), "reference_checked": sum(1 for k in got2 if str(k) in ref2),
                                             "reference_mismatches": sorted(k for k, v in got2.items() if str(k) in ref2 and ref2[str(k)] != v)}
        out[other]["voting"] = voting_stats(m2)
        # ---- mixed stream: the two workloads alternating through 3 pipelines (what a real image stream looks like to the
        #      history-dependent parts of the pipeline: assumed ring depth, voting launch budget); device-resident
        if not a.no_mixed_leg:
            out["mixed_stream"] = mixed_stream_leg(A, local_rank, W, H, D, m2, other, ids2[:6], a.workload, max(12, min(36, 2 * a.steps)))
        m2.release()
        # ---- the drop-in entry point with pageable host buffers
        out["host_inclusive"] = host_inclusive_leg(A, local_rank, W, H, D, a.workload, max(5, min(10, a.steps)))
        out["host_inclusive_registered"] = host_inclusive_leg(A, local_rank, W, H, D, a.workload, max(5, min(10, a.steps)), registered=True)
        out["host_farm"] = host_farm_leg(A, local_rank, W, H, D, a.workload, max(9, min(24, a.steps)))
        # ---- throughput mode: 3 pipelines in flight
        n3 = max(6, min(24, a.steps))
        ids3 = list(range(min(n3, 8)))
        m3, e3, t3, _, _ = measure_workload(A, local_rank, W, H, D, a.workload, n3, 3, 3, ids3)
        out["throughput_mode"] = {"value": round(t3 / e3, 4), "unit": "pairs/s", "in_flight_per_gpu": 3, "steps": n3,
                                  "note": "same workload, three pipelines (streams) in flight; per-kernel durations are then inflated by co-running kernels, which is why the headline region uses one"}
        m3.release()
        # ---- the same two legs on the OTHER workload (round-5 review: the natural-image numbers belong into this line)
        out[other]["host_inclusive"] = host_inclusive_leg(A, local_rank, W, H, D, other, max(5, min(10, a.steps)))
        n4 = max(6, min(18, a.steps))
        m4, e4, t4, _, _ = measure_workload(A, local_rank, W, H, D, other, n4, 3, 3, ids2[:6])
        out[other]["throughput_mode"] = {"value": round(t4 / e4, 4), "unit": "pairs/s", "in_flight_per_gpu": 3, "steps": n4,
                                         "async_fallbacks": m4.fallbacks}
        m4.release()
        # ---- THE DROP-IN FIGURE in one place: what a user of the reference gets from ADCensusStereo::Match(host, host, host)
        def brief(o):
            return {"value": o["value"], "unit": "pairs/s", "ms_per_pair": o["ms_per_pair"], "steps": o["steps"]}
        out["match_host"] = {"entry_point": "ADCensusStereo::Match == adc_match(left, right, disp): host images in, host map out, synchronous "
                                            "(upload of 12.4 MB, every kernel, download of 8.3 MB inside the timed call)",
                             a.workload: {"pageable": brief(out["host_inclusive"]), "registered": brief(out["host_inclusive_registered"])},
                             other: {"pageable": brief(out[other]["host_inclusive"])},
                             "note": "`value` of this line is adc_match_device (BASELINE.json: inputs resident in HBM when the timed region starts)"}
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(make_pair(a.workload, W, H, D, 0), D, a.cpu_rows, H)
            if a.cpu_baseline_structured and "structured" in out:
                out["structured"]["cpu_baseline"] = cpu_baseline(make_pair("structured", W, H, D, 0), D, a.cpu_rows, H)
            if a.cpu_all_cores > 1 and not a.no_extra_legs:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(a.workload, W, H, D, a.cpu_all_cores)
        if world == 1 and not a.no_cone_leg and not a.no_extra_legs:
            out["cone"] = cone_leg(A, local_rank, with_cpu=not a.no_cpu_baseline)
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)  # C-level buffers (RCCL / the reference) go where fd 1 points NOW: stderr
        except Exception:
            pass
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)  # (anything printed during teardown must not follow the JSON line)
    if dist is not None:
        dist.destroy_process_group()
    if backend is not None and "RCCL failed" in backend:
        sys.exit(3)  # (every rank: the communicator that was asked for did not carry the job)


def voting_stats(m):
    """Rounds (kernels of the chain that evaluated votes) and vote evaluations of the LAST Match of pipeline 0 -- the figures a
    slower box or another dispatch order of the voting chain would move first (round-5 advisor finding)."""
    try:
        r, e = m.handles[0].voting_stats()
        return {"rounds_last_match": r, "evaluations_last_match": e, "chain_budget_next": int(m.handles[0].debug_counter(3)),
                "band_to_xcd_sweep": int(m.handles[0].debug_counter(14))}
    except Exception:  # noqa: BLE001
        return None


def mixed_stream_leg(A, device, W, H, D, m_other, other, other_ids, workload, n):
    """`n` Matches alternating between the two workloads (pairs of `other` already resident in m_other's buffers, pairs 0..5 of
    `workload` uploaded here) through THREE pipelines in flight; every output compared with the reference table."""
    from adcensus_amd import farm
    ref = {w: farm.load_digest_tables(ROOT, w, [W, H, D])[0] or {} for w in (workload, other)}
    m = Matcher(A, device, W, H, D, 3)
    own_ids = list(range(min(6, max(1, len(other_ids)))))
    for pid in own_ids:
        l, r = make_pair(workload, W, H, D, pid)
        m.upload(("w", pid), l, r)
    for pid in other_ids:
        m.buf[("o", pid)] = m_other.buf[pid]  # (borrowed: freed by m_other)
    seq = []
    for k in range(n):
        seq.append(("o", other_ids[(k // 2) % len(other_ids)]) if k % 2 == 0 else ("w", own_ids[(k // 2) % len(own_ids)]))
    bad, checked = [], [0]

    def wait(slot):
        m.wait(slot)
        key = wait.pending.pop(slot)
        table = ref[other if key[0] == "o" else workload]
        if str(key[1]) in table:
            checked[0] += 1
            if farm.digest(m.output(key).tobytes()) != table[str(key[1])]:
                bad.append("%s %d" % (other if key[0] == "o" else workload, key[1]))
    wait.pending = {}

    def submit(slot, key):
        m.submit(slot, key)
        wait.pending[slot] = key
    farm.run_pairs(seq[:6], submit, wait, 3)  # warm-up: every pipeline has seen both kinds
    A.lib().adc_device_synchronize()
    t0 = time.perf_counter()
    farm.run_pairs(seq, m.submit, m.wait, 3)
    A.lib().adc_device_synchronize()
    dt = time.perf_counter() - t0
    farm.run_pairs(seq[:12], submit, wait, 3)  # untimed: outputs against the reference digests (the read-back costs 8 MB per pair)
    counters = {name: [int(st.debug_counter(c)) for st in m.handles] for name, c in
                (("aggregation_redos", 2), ("scanline_seam_redos", 4), ("voting_continuations", 1), ("median_handoff", 0),
                 ("plan_switches", 9), ("two_plan_matches", 10))}
    for key in list(m.buf):
        if key[0] == "o":
            m.buf.pop(key)
    m.release()
    return {"value": round(n / dt, 4), "unit": "pairs/s", "steps": n, "in_flight_per_gpu": 3,
            "workload": "alternating %s / %s pairs, %dx%d D=%d, device-resident" % (other, workload, W, H, D),
            "per_pipeline": counters, "reference_checked": checked[0], "reference_mismatches": bad,
            "note": "aggregation of a mixed stream: both ring plans are enqueued and the kernels choose on the device (no redo)"}


def host_inclusive_leg(A, device, W, H, D, workload, n, registered=False):
    left, right = make_pair(workload, W, H, D, 0)
    st = A.ADCensusStereo(device=device)
    if not st.Initialize(W, H, A.ADCensusOption(min_disparity=0, max_disparity=D)):
        raise SystemExit("Initialize failed: " + A.last_error())
    disp = np.empty((H, W), np.float32)  # pageable, like a caller's malloc'ed buffer
    if registered:  # opt-in: the caller page-locks its own buffers (adc_host_register) -> DMA without staging copies
        for arr in (left, right, disp):
            A.host_register(arr)
    for _ in range(2):
        assert st.Match(left, right, disp)
    t0 = time.perf_counter()
    for _ in range(n):
        assert st.Match(left, right, disp)
    dt = time.perf_counter() - t0
    st.Release()
    if registered:
        for arr in (left, right, disp):
            A.host_unregister(arr)
    return {"value": round(n / dt, 4), "unit": "pairs/s", "ms_per_pair": round(1000.0 * dt / n, 4), "steps": n,
            "entry_point": ("adc_match(left, right, disp) on buffers the caller registered with adc_host_register: H2D by DMA from the caller's "
                            "memory (12.4 MB), kernels, D2H into the caller's map (8.3 MB); synchronous") if registered else
                           ("adc_match(left, right, disp) == ADCensusStereo::Match with pageable host buffers: 2 host copies into pinned "
                            "staging (12.4 MB), H2D, kernels, D2H (8.3 MB), copy-out; synchronous")}


def cone_leg(A, device, with_cpu=True, reps=5):
    """BASELINE.json configs[0] / [1]: the Middlebury Cone pair (450 x 375, D = 64; tests/golden/cone_pair.npz = the reference's
    Data/Cone) through the drop-in entry point adc_match on one MI355X, >= 3 repetitions, and through the reference's own
    ADCensusStereo::Match on one host core, 3 repetitions (ADCensusStereo.cpp:69-132), the two maps compared bit for bit."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "cone_pair.npz"))
    left, right = np.ascontiguousarray(z["left"]), np.ascontiguousarray(z["right"])
    H, W, D = left.shape[0], left.shape[1], 64
    st = A.ADCensusStereo(device=device)
    if not st.Initialize(W, H, A.ADCensusOption(min_disparity=0, max_disparity=D)):
        raise SystemExit("Initialize failed: " + A.last_error())
    disp = np.empty((H, W), np.float32)
    for _ in range(3):
        assert st.Match(left, right, disp)
    ts = []
    for _ in range(max(3, reps)):
        t0 = time.perf_counter()
        assert st.Match(left, right, disp)
        ts.append(time.perf_counter() - t0)
    st.Release()
    out = {"workload": "Cone 450x375 D=64 (Middlebury, the reference's Data/Cone)", "gpu": {
        "entry_point": "adc_match(left, right, disp): pageable host buffers in and out, synchronous (PCIe inclusive)",
        "repetitions": len(ts), "ms_per_pair": [round(1000.0 * t, 4) for t in ts], "value": round(len(ts) / sum(ts), 3), "unit": "pairs/s"}}
    if with_cpu:
        from oracle import pyoracle  # checker / baseline leg only
        orc = pyoracle.load("auto")
        opt = pyoracle.Option(max_disparity=D)
        devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
        sys.stdout.flush()
        os.dup2(devnull, 1)  # (the reference printf()s its stage timings)
        secs, ref = [], None
        try:
            for _ in range(3):
                ref, s_ = orc.match(left, right, opt)
                secs.append(s_)
        finally:
            sys.stdout.flush()
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(saved, 1)
            os.close(devnull)
            os.close(saved)
        same = int((np.asarray(ref, np.float32).view(np.uint32) == disp.view(np.uint32)).sum())
        out["cpu"] = {"kind": "reference" if orc.kind == "reference" else "port", "cores": 1, "repetitions": 3,
                      "s_per_pair": [round(s_, 4) for s_ in secs], "value": round(3.0 / sum(secs), 4), "unit": "pairs/s",
                      "cpu_model": cpu_model(), "host_cores": os.cpu_count() or 0}
        out["check"] = {"pixels": int(disp.size), "bit_identical": same, "ok": same == int(disp.size)}
        out["gpu_over_cpu"] = round(out["gpu"]["value"] / out["cpu"]["value"], 1)
    return out


def host_farm_leg(A, device, W, H, D, workload, n):
    """The persistent farm of the C ABI (adc_farm_*): pageable host buffers in, pageable host buffers out, 3 pipelines."""
    pairs = [make_pair(workload, W, H, D, i) for i in range(min(n, 6))]
    farm = A.PairFarm(W, H, A.ADCensusOption(min_disparity=0, max_disparity=D), device=device, pipelines=3)
    outs = [np.empty((H, W), np.float32) for _ in range(3)]
    for i in range(3):
        farm.submit(pairs[i % len(pairs)][0], pairs[i % len(pairs)][1], outs[i % 3])
    farm.drain()
    t0 = time.perf_counter()
    for i in range(n):
        farm.submit(pairs[i % len(pairs)][0], pairs[i % len(pairs)][1], outs[i % 3])
    farm.drain()
    dt = time.perf_counter() - t0
    farm.close()
    return {"value": round(n / dt, 4), "unit": "pairs/s", "pipelines": 3, "steps": n,
            "entry_point": "adc_farm_submit / adc_farm_drain: pageable host images in, pageable host maps out, pinned staging ring inside"}


def pmc_traffic(workload, whd):
    """HBM bytes per aggregation launch from the committed rocprofv3 PMC passes (profiles/, FETCH_SIZE x2 + WRITE_SIZE,
    separate --pmc runs, gfx950 correction per MI355X_MICROARCH.md).  None when not collected for this workload / size
    or when it was collected on other aggregation kernel sources (k4_source_hash)."""
    tags = {(1920, 1080, 128): "", (1242, 375, 128): "_1242x375"}  # (the 1080p files carry no size suffix)
    if whd not in tags:
        return None
    for rnd in ("r6", "r5", "r4", "r3"):  # the newest measurement taken on exactly these kernel sources
        p = os.path.join(ROOT, "profiles", "%s_k4_pmc_traffic_%s%s.json" % (rnd, workload, tags[whd]))
        try:
            with open(p) as f:
                o = json.load(f)
            if o.get("k4_src_sha16") == k4_source_hash():
                return float(o["traffic_bytes_per_launch_avg"])
        except Exception:
            pass
    return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline_all_cores(workload, W, H, D, want):
    """SURVEY.md 8d, optional "all host cores" figure: N independent reference processes side by side, each matching its own
    pair (seed 12345 + i / 777 + i) once; value = N / wall time of the slowest.  N = min(want, cores, free memory / 2.6 GB)."""
    cores = os.cpu_count() or 1
    try:
        with open("/proc/meminfo") as f:
            avail_kb = [int(l.split()[1]) for l in f if l.startswith("MemAvailable")][0]
    except Exception:  # noqa: BLE001
        avail_kb = 8 << 20
    per_proc = 2.6e9 * (W * H * D) / (1920.0 * 1080 * 128) + 0.4e9
    n = int(max(1, min(want, cores, (0.6 * avail_kb * 1024.0) // per_proc)))
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "import bench\nfrom oracle import pyoracle\n"
            "orc = pyoracle.load('auto'); pair = bench.make_pair(%r, %d, %d, %d, int(sys.argv[1]))\n"
            "print('READY', flush=True); sys.stdin.readline()\n"
            "t0 = time.perf_counter(); orc.match(pair[0], pair[1], pyoracle.Option(max_disparity=%d)); print('SECS %%.3f' %% (time.perf_counter() - t0), flush=True)\n"
            % (ROOT, workload, W, H, D, D))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(i)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for i in range(n)]
    try:
        for pr in procs:  # every process has built its pair and loaded the oracle before the clock starts
            while True:
                line = pr.stdout.readline()
                if not line or line.startswith("READY"):
                    break
        t0 = time.perf_counter()
        for pr in procs:
            pr.stdin.write("go\n")
            pr.stdin.flush()
        secs = []
        for pr in procs:
            for line in pr.stdout:
                if line.startswith("SECS"):
                    secs.append(float(line.split()[1]))
                    break
        wall = time.perf_counter() - t0
    finally:
        for pr in procs:
            try:
                pr.stdin.close()
                pr.wait(timeout=30)
            except Exception:  # noqa: BLE001
                pr.kill()
    if len(secs) != n:
        return {"value": None, "error": "%d of %d reference processes finished" % (len(secs), n)}
    return {"value": round(n / wall, 5), "unit": "pairs/s", "cores": n, "processes": n, "host_cores": cores, "cpu_model": cpu_model(),
            "wall_s": round(wall, 2), "s_per_pair_per_process": [round(min(secs), 2), round(max(secs), 2)],
            "sample": "%d independent single-threaded reference processes side by side, one whole %s pair each (seeds differ), started together; "
                      "value = processes / wall time of the slowest" % (n, workload)}


def cpu_baseline(pair, D, rows, H):
    """Reference CPU path on a bounded sample: the top `rows` rows of the same pair, full width and
    disparity range, 1 thread (the reference is single-threaded).  pairs/s is scaled by rows/H."""
    from oracle import pyoracle  # checker / baseline leg only
    orc = pyoracle.load("auto")
    rows = H if rows <= 0 else min(rows, H)
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
            "cpu_model": cpu_model(), "host_cores": os.cpu_count() or 0, "build": getattr(orc, "build_info", "unknown"),
            "sample": ("the whole pair 0 of the batch (%d rows, full width, D=%d): %.2f s measured; the reference is single-threaded" % (H, D, secs)) if rows == H else
                      ("top %d of %d rows of pair 0 of the batch (full width, D=%d): %.2f s measured, scaled by rows to %.1f s/pair; "
                       "the reference is single-threaded" % (rows, H, D, secs, full_secs))}


if __name__ == "__main__":
    main()
