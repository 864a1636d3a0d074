#!/usr/bin/env python3
"""K8 with ordered sweeps of list chunks (analysis tool, CPU only; uses the oracle's dumps): kernels, votes and the longest sweep per
kernel when a wave owns C consecutive work-list entries and sweeps them in list order (tools/irv_chunk_sweep.cpp).  C = 1 is the
present chain.    python tools/irv_chunk_sweep.py [structured|noise] [W H D seed]
Time model per kernel: 4 us (boundary + entry / state / tile round trip) + max(longest sweep x 2.5 us, votes / 8192 waves x 5.4 us)
(a lone vote takes ~2.5 us, a vote of a full round ~5.4 us of its wave: profiles/r4_irv_chain_structured.txt)."""
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
    so = os.path.join(tempfile.mkdtemp(), "irv_chunk_sweep.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tools", "irv_chunk_sweep.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.irv_chunk_sweep.restype = C.c_long
    l, r = (workloads.structured_pair(W, H, D, seed=seed) if kind == "structured" else workloads.noise_pair(W, H, seed=seed))
    opt = pyoracle.Option(max_disparity=D)
    o = pyoracle.load("auto").run(l, r, opt)
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    arms, lab = np.ascontiguousarray(o["arms"]), np.ascontiguousarray(o["outlier_label"])
    print("%s %dx%d D=%d" % (kind, W, H, D))
    for ch in (1, 4, 8, 16, 32, 64):
        d = o["disp_after_lr"].copy()
        st = (C.c_long * (4 * 2000))()
        tot = lib.irv_chunk_sweep(P(d), P(lab), P(arms), W, H, opt.min_disparity, D, opt.irv_ts, C.c_float(opt.irv_th), ch, 1, st, 2000)
        ok = np.array_equal(d.view(np.uint32), o["disp_after_irv"].view(np.uint32))
        s = np.array(list(st), np.int64).reshape(-1, 4)[:tot]
        est = sum(4.0 + max(m * 2.5, v / 8192.0 * 5.4) for _, _, v, m in s)
        per = [int((s[:, 0] == p).sum()) for p in range(10)]
        print("  chunk %2d: %3d kernels %s  %.2f M votes  longest sweep per kernel p50 %d / p90 %d / max %d  estimated %.2f ms  equals the reference: %s"
              % (ch, tot, per, s[:, 2].sum() / 1e6, np.median(s[:, 3]), np.percentile(s[:, 3], 90), s[:, 3].max(), est / 1e3, ok))
        if ch in (1, 16):
            print("      pass 0: votes per kernel %s" % list(s[s[:, 0] == 0][:, 2]))
            print("      pass 0: longest sweep   %s" % list(s[s[:, 0] == 0][:, 3]))


if __name__ == "__main__":
    main()
