#!/usr/bin/env python3
"""Soak: N randomly drawn geometries / option sets, sized so that the median's column segments and speculative bands are ON (widths
384 .. 1400, heights 130 .. 460), whole Match x 2 against the CPU oracle, bit for bit; reports seam failures and fallbacks.
   python tools/gpu_soak_random.py [N [seed]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adcensus_amd as A  # noqa: E402
from adcensus_amd import workloads  # noqa: E402
from oracle import pyoracle  # noqa: E402
from tests import cases  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260930)
orc = pyoracle.load("auto")
bad, seam, segs = [], 0, {}
t0 = time.time()
for k in range(N):
    w, h = int(rng.integers(384, 1400)), int(rng.integers(130, 460))
    if rng.random() < 0.5:
        w &= ~1  # (even widths run the pair form with column segments; odd ones whole rows)
    d = int(rng.choice([16, 37, 64, 100, 128]))
    dmin = int(rng.choice([0, 0, 0, -9, 5]))
    kind = int(rng.integers(0, 3))
    sd = int(rng.integers(1, 1 << 30))
    pair = (workloads.structured_pair(w, h, d, seed=sd) if kind == 0 else
            workloads.quantized_noise_pair(w, h, d, seed=sd, levels=int(rng.choice([16, 32, 64]))) if kind == 1 else workloads.noise_pair(w, h, seed=sd))
    l1 = int(rng.choice([4, 9, 17, 34, 34]))
    opt = pyoracle.Option(min_disparity=dmin, max_disparity=dmin + d, cross_L1=l1, cross_L2=max(1, l1 // 2),
                          cross_t1=int(rng.integers(8, 40)), cross_t2=int(rng.integers(3, 12)),
                          irv_ts=int(rng.choice([0, 5, 20, 20, 45])), irv_th=float(rng.choice([0.1, 0.3, 0.4, 0.4, 0.7])),
                          lrcheck_thres=float(rng.choice([0.5, 1.0, 1.0, 2.0])))
    o = orc.run(pair[0], pair[1], opt, stages=["disp_final"])
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(w, h, cases.to_product_option(opt))
    for rep in range(2):
        got = st.match(*pair)
        if not np.array_equal(got.view(np.uint32), o["disp_final"].view(np.uint32)):
            bad.append((k, w, h, d, dmin, kind, rep, int((got.view(np.uint32) != o["disp_final"].view(np.uint32)).sum())))
    if int(st.debug_counter(7)):
        print('   seam failure: case %d %dx%d D=%d kind %d (%s)' % (k, w, h, d, kind, ('structured', 'quantized noise', 'noise')[kind]), flush=True)
    seam += int(st.debug_counter(7))
    segs[int(st.debug_counter(15))] = segs.get(int(st.debug_counter(15)), 0) + 1
    st.Release()
print("soak: %d cases in %.0f s, mismatching (case, w, h, D, dmin, kind, rep, pixels): %s; median seam failures %d; segments per band link of the last launch: %s"
      % (N, time.time() - t0, bad, seam, dict(sorted(segs.items()))))
sys.exit(1 if bad else 0)
