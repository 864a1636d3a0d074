#!/usr/bin/env python3
"""Timing experiment (GPU): stage times of three Matches of one 1080p pair -- used with libraries built with -DRR2_FAKE_MEM=1/2/3
(tools/build_variant.sh), in which the steady-state loads and / or stores of the full-ring aggregation passes go to one address per
wave: what is left of a pass's time when its memory streams cost nothing?  Results of such a library are garbage; only the times count.
    python tools/gpu_k4_fake_mem.py [structured|noise]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adcensus_amd as A  # noqa: E402
from adcensus_amd import workloads  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "structured"
    W, H, D = 1920, 1080, 128
    l, r = workloads.structured_pair(W, H, D, seed=777) if kind == "structured" else workloads.noise_pair(W, H, seed=12345)
    st = A.ADCensusStereo(device=0)
    opt = A.ADCensusOption()
    opt.max_disparity = D
    opt.do_lr_check = False  # (garbage volumes: keep the refinement out of it)
    opt.do_filling = False
    assert st.Initialize(W, H, opt)
    st.set_profiling(True)
    for i in range(4):
        st.match(l, r)
        ms = st.stage_ms()
        print("match %d  " % i + "  ".join("%s %.3f" % (k, v) for k, v in ms.items()) + "  | regular aggregation launch %.4f ms x %d" % st.aggregate_pass_ms(), flush=True)
    st.Release()


if __name__ == "__main__":
    main()
