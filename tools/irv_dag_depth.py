#!/usr/bin/env python3
"""Depth of the FINALITY dependency DAG of a region-voting pass (analysis tool, CPU only; uses the oracle's dumps).

A pure dataflow form of K8 ("evaluate an entry exactly once, when all its eligible predecessors are final") needs one
device-wide hand-off per LEVEL of this DAG: p depends on every eligible pixel of its cross region that precedes it in
raster order (multistep_refiner.cpp:173-196).  Measured on the SURVEY-8d structured pair (960x540, D=128):
    mismatch list  29 741 eligible pixels, DAG depth 4 325 (median level 1 337)
    occlusion list 23 865 eligible pixels, DAG depth 2 689
(a row of n adjacent invalid pixels is a chain of n, and every row below adds its left edge's right arm), against ~25
rounds per pass for the chaotic VALUE iteration the chain runs (values settle long before finality propagates).  At
1.3 us per cross-XCD hand-off the dataflow form would take 5-10 ms per pass at 1080p: not built.
    python tools/irv_dag_depth.py [W H D seed]
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
    W, H, D, seed = (int(v) for v in (sys.argv[1:5] + ["960", "540", "128", "777"][len(sys.argv) - 1:]))
    so = os.path.join(tempfile.mkdtemp(), "irv_dag_depth.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "irv_dag_depth.c"), "-o", so])
    lib = C.CDLL(so)
    l, r = workloads.structured_pair(W, H, D, seed=seed)
    o = pyoracle.load("auto").run(l, r, pyoracle.Option(max_disparity=D))
    arms = np.ascontiguousarray(o["arms"])
    lab, d = o["outlier_label"], o["disp_after_lr"]
    inv = d == np.float32(np.inf)
    for which, name in ((1, "mismatch"), (2, "occlusion")):
        el = np.ascontiguousarray(((lab == which) & inv).astype(np.uint8))
        dep = np.zeros((H, W), np.int32)
        e = C.c_long(0)
        m = lib.irv_depth(arms.ctypes.data_as(C.c_void_p), el.ctypes.data_as(C.c_void_p), W, H, dep.ctypes.data_as(C.c_void_p), C.byref(e))
        print("%-9s eligible %6d  DAG depth %5d  edges %8d  level p50/p90/p99 %s" % (name, int(el.sum()), m, e.value, np.percentile(dep[el > 0], [50, 90, 99]).astype(int)))


if __name__ == "__main__":
    main()
