#!/usr/bin/env python3
"""K5: how fast does the scanline recurrence forget its start?  (analysis tool, CPU only; uses the oracle's dumps)

A pass that starts in the middle of a path with the wrong state (the raw costs, like the reference's first pixel) is compared
with the full pass: steps until the whole state (all D values) is bit-identical.  If that number is small and bounded in
practice, a path can be cut into segments that start `warm-up` elements early and are VERIFIED against their predecessor's last
state (equal bits => the segment is exact; else it is redone): more chains than paths for the passes that are lone-wave chains.
    python tools/so_merge_length.py [noise|structured] [W H D seed]
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
    kind = sys.argv[1] if len(sys.argv) > 1 else "structured"
    a = sys.argv[2:6]
    W, H, D, seed = (int(v) for v in (a + ["480", "270", "128", "777" if kind == "structured" else "12345"][len(a):]))
    tmp = tempfile.mkdtemp()
    so = os.path.join(tmp, "so_merge.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(ROOT, "tools", "so_merge_length.cpp"), "-o", so])
    lib = C.CDLL(so)
    emul_so = os.path.join(ROOT, "tests", "emul", "_build", "libadcensus_emul.so")
    emul = C.CDLL(emul_so)
    l, r = (workloads.structured_pair(W, H, D, seed=seed) if kind == "structured" else workloads.noise_pair(W, H, seed=seed))
    opt = pyoracle.Option(max_disparity=D)
    o = pyoracle.load("auto").run(l, r, opt)
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    lh, lv, rh, rv = (np.zeros((H, W), np.uint8) for _ in range(4))
    emul.emul_color_diffs(P(l), P(lh), P(lv), W, H)
    emul.emul_color_diffs(P(r), P(rh), P(rv), W, H)
    a_, b_ = o["cost_aggr"].copy(), np.empty_like(o["cost_aggr"])
    print("%s %dx%d D=%d" % (kind, W, H, D))
    ML = 128
    for vert, dr, name in ((0, 1, "L->R"), (0, -1, "R->L"), (1, 1, "T->B"), (1, -1, "B->T")):
        emul.emul_scanline_pass(P(a_), P(b_), P(lv if vert else lh), P(rv if vert else rh), W, H, opt.min_disparity, D, vert, dr,
                                opt.so_tso, C.c_float(opt.so_p1), C.c_float(opt.so_p2))
        hist = (C.c_long * (ML + 1))()
        lib.so_merge(P(a_), P(b_), P(lv if vert else lh), P(rv if vert else rh), W, H, opt.min_disparity, D, vert, dr, opt.so_tso,
                     C.c_float(opt.so_p1), C.c_float(opt.so_p2), 37, ML, hist)
        h = np.array(list(hist), np.int64)
        n = h.sum()
        cum = np.cumsum(h)
        q = lambda f: int(np.searchsorted(cum, f * n))
        print("  pass %s: %d restarts; steps until the state is bit-identical: p50 %d, p90 %d, p99 %d, p99.9 %d, max %s (not merged within %d: %d)"
              % (name, n, q(0.5), q(0.9), q(0.99), q(0.999), (int(np.max(np.nonzero(h[:ML])[0])) if h[:ML].any() else "-"), ML, int(h[ML])))
        a_, b_ = b_, a_
    assert np.array_equal(a_.view(np.uint32), o["cost_so"].view(np.uint32))


if __name__ == "__main__":
    main()
