"""Rounds / vote evaluations / chain budget of the region-voting chain on one structured pair (ADC_IRV_SLACK from the environment).
   python tools/gpu_k8_stats.py [W H [seed]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import adcensus_amd as A
from adcensus_amd import workloads

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 777
D = 128
left, right = workloads.structured_pair(W, H, D, seed=seed)
st = A.ADCensusStereo(device=0)
assert st.Initialize(W, H, A.ADCensusOption(max_disparity=D))
st.set_profiling(True)
out = np.empty((H, W), np.float32)
for i in range(4):
    assert st.Match(left, right, out)
t0 = time.perf_counter()
for i in range(5):
    assert st.Match(left, right, out)
dt = (time.perf_counter() - t0) / 5
r, e = st.voting_stats()
print("slack=%s %dx%d seed %d: rounds %d evaluations %d budget %d continuations %d refine %.3f ms match %.3f ms" % (
    os.environ.get("ADC_IRV_SLACK", "1"), W, H, seed, r, e, st.debug_counter(3), st.debug_counter(1), st.stage_ms()["refine"], dt * 1e3))
