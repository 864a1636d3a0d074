#!/usr/bin/env python3
"""Timing probe of the banded median (K11): wall time of adc_debug_run(RUN_MEDIAN) for image heights that give 1, 2, 4, 9, 17
bands at W = 1920 (and a narrow image), 30 runs each, minimum -- separates the per-level cost of ONE band from the band-to-band
lag.  python tools/gpu_median_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adcensus_amd as A  # noqa: E402

for (W, H) in ((1920, 64), (1920, 128), (1920, 256), (1920, 576), (1920, 1080), (480, 1080), (3840, 64)):
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(W, H, A.ADCensusOption(max_disparity=16))
    rng = np.random.default_rng(1)
    d = rng.uniform(0, 15, (H, W)).astype(np.float32)
    best = 1e9
    for _ in range(30):
        st.debug_write(A.BUF_DISP_LEFT, d)
        t0 = time.perf_counter()
        st.debug_run(A.RUN_MEDIAN)
        best = min(best, time.perf_counter() - t0)
    levels = W + 2 * (H - 1)
    print("W %4d H %4d bands %2d levels %5d  %.1f us  -> %.1f ns per level" % (W, H, (H + 63) // 64, levels, best * 1e6, best * 1e9 / levels))
    st.Release()
