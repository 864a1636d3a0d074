"""Chain lengths (rounds) of the region voting over a stream of DISTINCT structured pairs (seeds 777 + i), one handle, in batch order --
the data behind the launch budget's margin (k_voting.hip: adc_voting_finish).   python tools/gpu_k8_chain_lengths.py [W H [pairs [cycles]]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import adcensus_amd as A
from adcensus_amd import workloads

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 10
CY = int(sys.argv[4]) if len(sys.argv) > 4 else 3
D = 128
pairs = [workloads.structured_pair(W, H, D, seed=777 + i) for i in range(NP)]
st = A.ADCensusStereo(device=0)
assert st.Initialize(W, H, A.ADCensusOption(max_disparity=D))
out = np.empty((H, W), np.float32)
rounds, budgets, conts = [], [], []
for c in range(CY):
    for l, r in pairs:
        b = st.debug_counter(3)
        c0 = st.debug_counter(1)
        assert st.Match(l, r, out)
        rounds.append(st.voting_stats()[0])
        budgets.append(b)
        conts.append(st.debug_counter(1) - c0)
print("%dx%d: rounds per Match %s" % (W, H, rounds))
print("budget in force %s" % budgets)
print("continuations %s" % conts)
used = [r + 4 for r in rounds]  # BEGIN, BEGIN2, FINAL + the kernel that finds the chain finished
for margin in (0.40, 0.25, 0.15, 0.10):
    hist, over, surplus = [0] * 8, 0, 0
    budget = 256
    for i, u in enumerate(used):
        if u > budget:
            over += 1
        else:
            surplus += budget - u
        hist[i & 7] = u
        longest = max(hist)
        budget = longest + max(int(longest * margin), 2) + 2
    print("margin %.2f: overruns %d of %d, surplus kernels per Match %.1f" % (margin, over, len(used), surplus / float(len(used))))
