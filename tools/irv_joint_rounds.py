#!/usr/bin/env python3
"""K8 as ONE fixed-point iteration over all ten passes (analysis tool, CPU only; uses the oracle's dumps): rounds and vote
evaluations of the joint Jacobi iteration (tools/irv_joint_rounds.cpp) against the per-pass rounds of the present chain.
    python tools/irv_joint_rounds.py [structured|noise] [W H D seed]
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
    tmp = tempfile.mkdtemp()
    so = os.path.join(tmp, "irv_joint.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tools", "irv_joint_rounds.cpp"), "-o", so])
    so2 = os.path.join(tmp, "irv_local.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tools", "irv_local_rounds.cpp"), "-o", so2])
    lib, lib2 = C.CDLL(so), C.CDLL(so2)
    lib.irv_joint.restype = C.c_long
    lib2.irv_local.restype = C.c_long
    l, r = (workloads.structured_pair(W, H, D, seed=seed) if kind == "structured" else workloads.noise_pair(W, H, seed=seed))
    opt = pyoracle.Option(max_disparity=D)
    o = pyoracle.load("auto").run(l, r, opt, stages=["arms", "outlier_label", "disp_after_lr", "disp_after_irv"])
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    arms, lab, d0 = np.ascontiguousarray(o["arms"]), np.ascontiguousarray(o["outlier_label"]), np.ascontiguousarray(o["disp_after_lr"])
    print("%s %dx%d D=%d seed %d: %d mismatches, %d occlusions" % (kind, W, H, D, seed, int((lab == 1).sum()), int((lab == 2).sum())))
    d = d0.copy()
    per = (C.c_long * 10)()
    ev = C.c_long(0)
    tot = lib2.irv_local(P(d), P(lab), P(arms), W, H, opt.min_disparity, D, opt.irv_ts, C.c_float(opt.irv_th), 64, 1, 1, 1, per, C.byref(ev))
    print("  present chain (one Jacobi round per kernel, pass after pass): %d rounds %s, equals the reference: %s"
          % (tot, list(per), np.array_equal(d.view(np.uint32), o["disp_after_irv"].view(np.uint32))))
    out = np.empty_like(d0)
    MR = 4096
    evr, chr_ = (C.c_long * MR)(), (C.c_long * MR)()
    rounds = lib.irv_joint(P(out), P(d0), P(lab), P(arms), W, H, opt.min_disparity, D, opt.irv_ts, C.c_float(opt.irv_th), 10, evr, chr_, MR)
    ok = np.array_equal(out.view(np.uint32), o["disp_after_irv"].view(np.uint32))
    e = np.array(list(evr))[:rounds]
    c = np.array(list(chr_))[:rounds]
    print("  joint iteration over all 10 levels: %d rounds, equals the reference: %s" % (rounds, ok))
    print("    evaluations (per-entry dirty test): total %.2f M; per round %s" % (e.sum() / 1e6, [int(v) for v in e]))
    print("    changes per round %s" % [int(v) for v in c])
    lib.irv_joint_px.restype = C.c_long
    for jac, dm, what in ((1, 0, "Jacobi, exact dirty test"), (1, 1, "Jacobi, 8x8 change tiles over the region's bounding box"),
                          (1, 2, "Jacobi, 8x8 change tiles that remember the lowest fill iteration of their changes"),
                          (0, 1, "in place (raster order), 8x8 change tiles")):
        votes = C.c_long(0)
        rounds = lib.irv_joint_px(P(out), P(d0), P(lab), P(arms), W, H, opt.min_disparity, D, opt.irv_ts, C.c_float(opt.irv_th), jac, dm, evr, chr_,
                                  C.byref(votes), MR)
        ok = np.array_equal(out.view(np.uint32), o["disp_after_irv"].view(np.uint32))
        e = np.array(list(evr))[:rounds]
        c = np.array(list(chr_))[:rounds]
        print("  one state (fill iteration, bin) per pixel, %s: %d rounds, equals the reference: %s" % (what, rounds, ok))
        print("    pixel evaluations %.3f M (%.3f M vote decisions); per round %s" % (e.sum() / 1e6, votes.value / 1e6, [int(v) for v in e]))
        print("    changes per round %s" % [int(v) for v in c])

    lib.irv_joint_chunks.restype = C.c_long
    stp = (C.c_long * MR)()
    for Cn, rm in ((8, 0), (16, 0), (32, 0), (64, 0), (32, 1)):
        rounds = lib.irv_joint_chunks(P(out), P(d0), P(lab), P(arms), W, H, opt.min_disparity, D, opt.irv_ts, C.c_float(opt.irv_th), Cn, rm, evr, chr_, stp, MR)
        ok = np.array_equal(out.view(np.uint32), o["disp_after_irv"].view(np.uint32))
        e = np.array(list(evr))[:rounds]
        print("  %s-major chunks of %d entries per wave, evaluated in order, in place: %d rounds, equals the reference: %s; evaluations %.3f M %s; "
              "sequential steps per round %s" % ("row" if rm else "column", Cn, rounds, ok, e.sum() / 1e6, [int(v) for v in e], [int(v) for v in list(stp)[:rounds]]))

    lib.irv_joint_bands.restype = C.c_long
    for R, la in ((8, 0), (16, 0), (32, 0), (64, 0), (135, 0), (H, 0), (8, 1), (16, 1)):
        rounds = lib.irv_joint_bands(P(out), P(d0), P(lab), P(arms), W, H, opt.min_disparity, D, opt.irv_ts, C.c_float(opt.irv_th), R, la, evr, chr_, MR)
        ok = np.array_equal(out.view(np.uint32), o["disp_after_irv"].view(np.uint32))
        e = np.array(list(evr))[:rounds]
        print("  bands of %d rows swept row by row, all bands side by side%s: %d rounds, equals the reference: %s; evaluations %.3f M %s"
              % (R, ", iteration-aware dirty test" if la else "", rounds, ok, e.sum() / 1e6, [int(v) for v in e]))


if __name__ == "__main__":
    main()
