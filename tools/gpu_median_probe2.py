#!/usr/bin/env python3
"""Timing probe of the banded median with column segments (K11, round 6): wall time of adc_debug_run(RUN_MEDIAN) at 1920x1080 for
segment counts / warm-up widths (own interpreter per variant: the switches are read once), minimum of 40 runs.
python tools/gpu_median_probe2.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, time
sys.path.insert(0, %r)
import numpy as np
import adcensus_amd as A
W, H = %d, %d
st = A.ADCensusStereo(device=0)
assert st.Initialize(W, H, A.ADCensusOption(max_disparity=16))
rng = np.random.default_rng(1)
d = np.floor(rng.uniform(0, 15, (H, W))).astype(np.float32)
best = 1e9
for _ in range(40):
    st.debug_write(A.BUF_DISP_LEFT, d)
    t0 = time.perf_counter()
    st.debug_run(A.RUN_MEDIAN)
    best = min(best, time.perf_counter() - t0)
print("%%.1f us  seam failures %%d" %% (best * 1e6, st.debug_counter(7)))
"""
for (W, H) in ((1920, 1080), (1920, 192)):
    for env in ({"ADC_MEDIAN_SEG": "1"}, {"ADC_MEDIAN_SEG": "2"}, {"ADC_MEDIAN_SEG": "4"}, {"ADC_MEDIAN_SEG": "8"},
                {"ADC_MEDIAN_SEG": "8", "ADC_MEDIAN_WARM": "16"}, {"ADC_MEDIAN_SEG": "8", "ADC_MEDIAN_WARM": "64"}, {"ADC_MEDIAN_SEG": "8", "ADC_MEDIAN_WARM": "256"},
                {"ADC_MEDIAN_SEG": "8", "ADC_MEDIAN_WARM": "512"}, {"ADC_MEDIAN_SEG": "8", "ADC_MEDIAN_SPEC": "1"}, {"ADC_MEDIAN_SEG": "8", "ADC_MEDIAN_SPEC": "3"}):
        out = subprocess.run([sys.executable, "-c", CODE % (ROOT, W, H)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        print("%dx%d %-60s %s" % (W, H, env, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
