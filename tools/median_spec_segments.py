#!/usr/bin/env python3
"""K11 with speculative column segments (analysis tool, CPU only; uses the oracle's dumps; tools/median_spec_segments.c): does a
segment of a band that starts `warm` columns to the left of its first column, with that edge treated like the image border (and, as
before, `run-in` rows above its band from the raw row above), arrive at the true filter's state?
    python tools/median_spec_segments.py noise,1920,1080,12345 structured,1920,1080,777 ...
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
    so = os.path.join(tempfile.mkdtemp(), "median_spec_segments.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "median_spec_segments.c"), "-o", so])
    lib = C.CDLL(so)
    lib.spec_segments.restype = C.c_long
    orc = pyoracle.load("auto")
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    for arg in sys.argv[1:] or ["noise,960,540,12345", "structured,960,540,777"]:
        kind, W, H, seed = arg.split(",")
        W, H, seed, D = int(W), int(H), int(seed), 128
        l, r = workloads.structured_pair(W, H, D, seed=seed) if kind == "structured" else workloads.noise_pair(W, H, seed=seed)
        o = orc.run(l, r, pyoracle.Option(max_disparity=D), stages=["disp_after_interp", "disp_final"])
        raw, fin = np.ascontiguousarray(o["disp_after_interp"]), np.ascontiguousarray(o["disp_final"])
        for nseg in (4, 8):
            for warm in (64, 96, 128, 192):
                hb, vb, pairs = C.c_long(0), C.c_long(0), C.c_long(0)
                bad = lib.spec_segments(P(raw), P(fin), W, H, 64, nseg, 128, warm, C.byref(hb), C.byref(vb), C.byref(pairs))
                print("%s %dx%d seed %d: %d segments, warm-up %3d columns (run-in 128 rows): %d of %d (band, segment) pairs fail (column seams %d, row seams %d)"
                      % (kind, W, H, seed, nseg, warm, bad, pairs.value, hb.value, vb.value), flush=True)


if __name__ == "__main__":
    main()
