"""Stress of the asynchronous pipeline's assumptions (tools/, not a test: ~1 GPU-minute): three pipelines in flight on one
GPU fed with an alternating sequence of DIFFERENT structured and noise pairs (1080p, D = 128), so that the assumed ring
depth, the voting launch budget and the co-residency of the median bands are all wrong or disturbed again and again.
Every output is compared with the output of the same pair computed alone on a fresh handle; the fallback counters
(median hand-off fallbacks, voting continuations, aggregation redos) are printed.
  python tools/gpu_stress_mixed.py [rounds]"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import adcensus_amd as A
from adcensus_amd import workloads

W, H, D = 1920, 1080, 128
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pairs = []
for i in range(3):
    pairs.append(("structured%d" % i, workloads.structured_pair(W, H, D, seed=900 + i)))
    pairs.append(("noise%d" % i, workloads.noise_pair(W, H, 4242 + i)))
opt = A.ADCensusOption(min_disparity=0, max_disparity=D)
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
# reference outputs: every pair alone on a fresh handle (second Match of the handle: budgets adapted)
want = {}
for name, (l, r) in pairs:
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(W, H, opt)
    a = st.match(l, r); b = st.match(l, r)
    assert sha(a) == sha(b), name + ": not repeatable"
    want[name] = sha(b)
    st.Release()
print("solo outputs done", flush=True)
hs = []
for _ in range(3):
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(W, H, opt)
    hs.append(st)
outs = [np.empty((H, W), np.float32) for _ in range(3)]
t0 = time.perf_counter()
seq = [pairs[(3 * k + (k // 7)) % len(pairs)] for k in range(rounds * len(pairs))]
bad = 0
inflight = []
for k, (name, (l, r)) in enumerate(seq):
    slot = k % 3
    if len(inflight) == 3:
        s0, n0 = inflight.pop(0)
        assert hs[s0].wait()
        if sha(outs[s0]) != want[n0]:
            bad += 1; print("MISMATCH", n0, flush=True)
    assert hs[slot].match_async(l, r, outs[slot])
    inflight.append((slot, name))
for s0, n0 in inflight:
    assert hs[s0].wait()
    if sha(outs[s0]) != want[n0]:
        bad += 1; print("MISMATCH", n0, flush=True)
dt = time.perf_counter() - t0
print("pairs %d in %.2f s (%.1f pairs/s, host buffers, 3 in flight, alternating structured / noise), mismatches %d" % (len(seq), dt, len(seq) / dt, bad))
print("per pipeline [median fallbacks, voting continuations, aggregation redos, next voting budget, scanline seam redos, median seam failures, "
      "plan switches, two-plan Matches, partial redos]:",
      [[int(st.debug_counter(c)) for c in (0, 1, 2, 3, 4, 7, 9, 10, 11)] for st in hs])
for st in hs:
    st.Release()
sys.exit(1 if bad else 0)
