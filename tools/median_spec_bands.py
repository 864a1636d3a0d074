#!/usr/bin/env python3
"""K11 with speculative bands (analysis tool, CPU only; uses the oracle's dumps): does a band of the in-place 3x3 median that
starts from the UNFILTERED row above it reproduce, `run-in` rows further down, the rows of the true recursive filter?
For every band seam (multiples of 64 rows) the rows [seam - run_in, seam) are filtered in place starting from raw rows above
and the last of them is compared bit for bit with the reference's.  k_median_banded (spec = 1) relies on run_in = 64.
    python tools/median_spec_bands.py noise,1920,1080,12345 structured,1920,1080,777 ...
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from adcensus_amd import workloads
    from oracle import pyoracle
    so = os.path.join(tempfile.mkdtemp(), "median_spec_bands.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "median_spec_bands.c"), "-o", so])
    lib = C.CDLL(so)
    lib.spec_bands.restype = C.c_long
    orc = pyoracle.load("auto")
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    for arg in sys.argv[1:] or ["noise,960,540,12345", "structured,960,540,777"]:
        kind, W, H, seed = arg.split(",")
        W, H, seed, D = int(W), int(H), int(seed), 128
        l, r = workloads.structured_pair(W, H, D, seed=seed) if kind == "structured" else workloads.noise_pair(W, H, seed=seed)
        o = orc.run(l, r, pyoracle.Option(max_disparity=D), stages=["disp_after_interp", "disp_final"])
        raw, fin = np.ascontiguousarray(o["disp_after_interp"]), np.ascontiguousarray(o["disp_final"])
        for run_in in (8, 16, 32, 64, 128):
            worst, seams = C.c_long(0), C.c_long(0)
            bad = lib.spec_bands(P(raw), P(fin), W, H, 64, run_in, C.byref(worst), C.byref(seams))
            print("%s %dx%d seed %d: run-in %3d rows: %d of %d seams differ (worst: %d pixels of the seam row)"
                  % (kind, W, H, seed, run_in, bad, seams.value, worst.value), flush=True)


if __name__ == "__main__":
    main()
