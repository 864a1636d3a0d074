#!/usr/bin/env python3
"""Ray-walk statistics of the interpolation stage K9 (analysis tool, CPU only; uses the oracle's dumps).

For both target lists of a pair (mismatches on the map after region voting, occlusions on the map after the first list's
fills) the 16 rays of every target are walked like k_interpolate_tab walks them, for several (first trip, later trips) step
counts and two caps of the cell-distance map, counting requested map values (gathers), ray round trips and WAVE round trips
(4 targets x 16 rays per wave; a wave iterates until its last ray has ended).
    python tools/itp_ray_stats.py [noise|structured] [W H D seed]
"""
import ctypes as C
import math
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ray_table(max_search):
    pi = float(np.float32(3.1415926))
    tab = np.zeros((max_search, 16), np.int32)
    ang = 0.0
    for s in range(16):
        sa, ca = math.sin(ang), math.cos(ang)
        for m in range(1, max_search):
            dy, dx = int(math.floor(m * sa + 0.5)) if m * sa >= 0 else -int(math.floor(-m * sa + 0.5)), \
                     int(math.floor(m * ca + 0.5)) if m * ca >= 0 else -int(math.floor(-m * ca + 0.5))
            tab[m, s] = (dy << 16) | (dx & 0xffff)
        ang += pi / 16
    return tab


def main():
    from adcensus_amd import workloads
    from oracle import pyoracle
    kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
    W, H, D, seed = (int(v) for v in (sys.argv[2:6] + ["960", "540", "128", "12345"][len(sys.argv) - 2 if len(sys.argv) > 2 else 0:]))
    so = os.path.join(tempfile.mkdtemp(), "itp_ray_stats.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "itp_ray_stats.c"), "-o", so, "-lm"])
    lib = C.CDLL(so)
    l, r = (workloads.noise_pair(W, H, seed=seed) if kind == "noise" else workloads.structured_pair(W, H, D, seed=seed))
    o = pyoracle.load("auto").run(l, r, pyoracle.Option(max_disparity=D))
    lab = o["outlier_label"]
    inv = np.float32(np.inf)
    tab = ray_table(D)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    dmap = o["disp_after_irv"].copy()
    print("%s %dx%d D=%d: %d pixels" % (kind, W, H, D, W * H))
    for which, name in ((1, "mismatch"), (2, "occlusion")):
        targets = np.ascontiguousarray(np.flatnonzero(((lab == which) & (dmap == inv)).ravel()).astype(np.int32))
        valid = np.ascontiguousarray((dmap != inv).astype(np.uint8))
        n = len(targets)
        print("  %-9s targets %8d (%.1f %% of the image)" % (name, n, 100.0 * n / (W * H)))
        base = None
        for cap in (16, 64):
            cdist = np.zeros(((H + 1) // 2, (W + 1) // 2), np.uint8)
            lib.itp_cdist(P(valid), W, H, cap, P(cdist))
            for ns0, ns in ((4, 4), (2, 4), (2, 2), (1, 4), (8, 8)):
                g, t, wt = C.c_longlong(), C.c_longlong(), C.c_longlong()
                hist = np.zeros(64, np.int64)
                lib.itp_walk(P(valid), P(cdist), W, H, P(tab), D, P(targets), C.c_long(n), ns0, ns, C.byref(g), C.byref(t), C.byref(wt), P(hist))
                if base is None:
                    base = (g.value, wt.value)
                print("    cap %2d  steps/trip %d then %d: gathers/target %6.1f (%.2fx)  ray trips/target %5.1f  wave trips/group %5.2f (%.2fx)  rays ending in trip 1 / 2 / 3+: %.0f / %.0f / %.0f %%"
                      % (cap, ns0, ns, g.value / max(n, 1), g.value / max(base[0], 1), t.value / max(n, 1), wt.value / max((n + 3) // 4, 1),
                         wt.value / max(base[1], 1), 100.0 * hist[1] / max(hist.sum(), 1), 100.0 * hist[2] / max(hist.sum(), 1),
                         100.0 * hist[3:].sum() / max(hist.sum(), 1)))
        # the fills of this list are written back before the next list is walked (multistep_refiner.cpp:298-303)
        fin = o["disp_after_interp"]
        dmap.ravel()[targets] = fin.ravel()[targets]


if __name__ == "__main__":
    main()
