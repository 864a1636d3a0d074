#!/usr/bin/env python3
"""K8: trips of the per-pixel atomic loop of a vote, now and with the first (and second) bin of a block merged into one atomic
(analysis tool, CPU only; oracle dumps; tools/irv_block_stats.cpp).    python tools/irv_block_stats.py [structured|noise] [W H D seed]"""
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
    kind = sys.argv[1] if len(sys.argv) > 1 else "structured"
    a = sys.argv[2:6]
    W, H, D, seed = (int(v) for v in (a + ["960", "540", "128", "777" if kind == "structured" else "12345"][len(a):]))
    so = os.path.join(tempfile.mkdtemp(), "irv_block_stats.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tools", "irv_block_stats.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.irv_block_stats.restype = C.c_long
    l, r = (workloads.structured_pair(W, H, D, seed=seed) if kind == "structured" else workloads.noise_pair(W, H, seed=seed))
    opt = pyoracle.Option(max_disparity=D)
    o = pyoracle.load("auto").run(l, r, opt)
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    arms, lab = np.ascontiguousarray(o["arms"]), np.ascontiguousarray(o["outlier_label"])
    d = np.ascontiguousarray(o["disp_after_lr"])
    print("%s %dx%d D=%d (state at the start of the first pass of each list)" % (kind, W, H, D))
    for which, name in ((1, "label 1"), (2, "label 2")):
        out = (C.c_double * 8)()
        n = lib.irv_block_stats(P(d), P(lab), P(arms), W, H, opt.min_disparity, D, which, out)
        if not n:
            continue
        v = list(out)
        print("  %s: %d votes, %.0f %% with a block of several bins (%.1f %% of the %.0f blocks); trips of the per-pixel loop per vote: now %.2f, "
              "first bin merged %.2f, first two bins merged %.2f" % (name, n, 100 * v[1] / v[0], 100 * v[6] / max(v[5], 1), v[5], v[2] / v[0], v[3] / v[0], v[4] / v[0]))


if __name__ == "__main__":
    main()
