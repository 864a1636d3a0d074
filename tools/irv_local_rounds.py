#!/usr/bin/env python3
"""K8 with tile-local rounds (analysis tool, CPU only; uses the oracle's dumps): how many kernels does a region-voting pass need
when every tile runs up to T rounds on a snapshot before it writes back?  T = 1 is the present chain (one round per kernel).
    python tools/irv_local_rounds.py [structured|noise] [W H D seed]
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
    W, H, D, seed = (int(v) for v in (a + ["960", "540", "128", "777" if kind == "structured" else "12345"][len(a):]))
    so = os.path.join(tempfile.mkdtemp(), "irv_local_rounds.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tools", "irv_local_rounds.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.irv_local.restype = C.c_long
    l, r = (workloads.structured_pair(W, H, D, seed=seed) if kind == "structured" else workloads.noise_pair(W, H, seed=seed))
    opt = pyoracle.Option(max_disparity=D)
    o = pyoracle.load("auto").run(l, r, opt)
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    arms, lab = np.ascontiguousarray(o["arms"]), np.ascontiguousarray(o["outlier_label"])
    print("%s %dx%d D=%d" % (kind, W, H, D))
    inv = o["disp_after_lr"] == np.float32(np.inf)
    for S in (32, 64):
        el = ((lab > 0) & inv).astype(np.int64)
        hh, ww = (H + S - 1) // S * S, (W + S - 1) // S * S
        pad = np.zeros((hh, ww), np.int64)
        pad[:H, :W] = el
        per_tile = pad.reshape(hh // S, S, ww // S, S).sum(axis=(1, 3)).ravel()
        print("  eligible pixels of the first two passes per %dx%d tile: mean %.0f, p50 %d, p90 %d, p99 %d, max %d (of %d tiles, %d empty)"
              % (S, S, per_tile.mean(), *np.percentile(per_tile, [50, 90, 99]).astype(int), per_tile.max(), len(per_tile), int((per_tile == 0).sum())))
    for S, T, J in ((64, 1, 1), (64, 2, 1), (64, 4, 1), (64, 8, 1), (64, 16, 1), (64, 64, 1), (128, 16, 1), (32, 4, 1), (32, 8, 1), (32, 16, 1), (64, 1, 0)):
        d = o["disp_after_lr"].copy()
        per = (C.c_long * 10)()
        ev = C.c_long(0)
        tot = lib.irv_local(P(d), P(lab), P(arms), W, H, opt.min_disparity, D, opt.irv_ts, C.c_float(opt.irv_th), S, T, J, 1, per, C.byref(ev))
        ok = np.array_equal(d.view(np.uint32), o["disp_after_irv"].view(np.uint32))
        print("  tiles %3dx%-3d  %s, up to %2d local rounds: %3d kernels over the 10 passes %s  %.2f M vote evaluations  equals the reference: %s"
              % (S, S, "parallel rounds" if J else "sequential sweep", T, tot, list(per), ev.value / 1e6, ok))
    # the same with tile skipping and a per-entry dirty test, per-kernel statistics and a time estimate: a kernel costs its
    # boundary (2.5 us) + the busiest tile's votes shared by the 16 waves of a workgroup at ~1.5 us per vote and wave (LDS-resident
    # state) -- or, when more tiles are active than the chip holds workgroups (512), the total votes over all 4096 wave slots
    lib.irv_local_stats.restype = C.c_long
    for S, T in ((64, 8), (32, 8), (64, 16)):
        d = o["disp_after_lr"].copy()
        st = (C.c_long * (4 * 400))()
        tot = lib.irv_local_stats(P(d), P(lab), P(arms), W, H, opt.min_disparity, D, opt.irv_ts, C.c_float(opt.irv_th), S, T, 1, st, 400)
        ok = np.array_equal(d.view(np.uint32), o["disp_after_irv"].view(np.uint32))
        a = np.array(list(st), np.int64).reshape(-1, 4)[:tot]
        est = sum(2.5 + max(m / 16.0, v / 4096.0) * 1.5 for _, _, v, m in a)
        print("  tiles %dx%d, up to %d local rounds, tile skipping + dirty test: %d kernels, %.2f M votes, busiest tile per kernel p50 %d / max %d votes, "
              "active tiles per kernel p50 %d; equals the reference: %s; estimated %.2f ms at this size"
              % (S, S, T, tot, a[:, 2].sum() / 1e6, np.median(a[:, 3]), a[:, 3].max(), np.median(a[:, 1]), ok, est / 1e3))
        first = a[[i for i in range(len(a)) if i == 0 or a[i, 0] != a[i - 1, 0]]]
        print("      first kernel of a pass: votes %s, busiest tile %s" % (list(first[:, 2]), list(first[:, 3])))


if __name__ == "__main__":
    main()
