#!/usr/bin/env python3
"""GPU-box diagnostic: per-stage parity report for a list of cases + stage timings of the big
configs.  Writes JSON lines to gpurun_out/diag.jsonl as it goes (survives a later crash)."""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import adcensus_amd as A  # noqa: E402
from adcensus_amd import workloads  # noqa: E402
from oracle import pyoracle  # noqa: E402
from tests import cases, gpu_harness  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
log = open(os.path.join(OUT, "diag.jsonl"), "a")


def emit(obj):
    log.write(json.dumps(obj) + "\n")
    log.flush()
    print(json.dumps(obj)[:600], flush=True)


def parity(names):
    orc = pyoracle.load("auto")
    for name in names:
        try:
            left, right, opt = cases.make_case(name)
            o = orc.run(left, right, opt)
            t = time.time()
            rep = gpu_harness.stage_report(left, right, opt, o)
            emit({"case": name, "oracle": orc.kind, "secs": round(time.time() - t, 2),
                  "failing": gpu_harness.failing(rep), "stages_ok": [k for k, v in rep.items() if not v["bad"]],
                  "voting": rep.get("disp_after_irv", {}).get("voting_rounds_evals")})
        except Exception as e:  # noqa
            emit({"case": name, "error": repr(e), "tb": traceback.format_exc()[-1500:]})


def timing(label, left, right, D, reps=5):
    h, w = left.shape[:2]
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(w, h, A.ADCensusOption(max_disparity=D)), A.last_error()
    st.set_profiling(True)
    d = np.empty((h, w), np.float32)
    rows = []
    for i in range(reps):
        t = time.perf_counter()
        ok = st.Match(left, right, d)
        wall = (time.perf_counter() - t) * 1e3
        rows.append({"wall_ms": round(wall, 3), "ok": ok, "stage_ms": {k: round(v, 3) for k, v in st.stage_ms().items()},
                     "agg_pass_ms": st.aggregate_pass_ms(), "voting": st.voting_stats()})
    emit({"timing": label, "shape": [h, w, D], "runs": rows, "inf": int(np.isinf(d).sum())})
    st.Release()


if __name__ == "__main__":
    what = sys.argv[1:] or ["parity", "timing"]
    emit({"version": A.lib().adc_version().decode(), "devices": A.device_count()})
    if "parity" in what:
        parity(["q_3x3_d2", "q_20x40_d32", "s2_96x64_d32", "q_257x131_d64", "s2_150x100_neg", "s2_320x180_d128",
                "s2_200x120_d200", "cone"])
    if "timing" in what:
        l, r = cases.cone_pair()
        timing("cone", l, r, 64)
        l, r = workloads.noise_pair(1920, 1080, 12345)
        timing("noise1080", l, r, 128, reps=4)
        l, r = workloads.structured_pair(1920, 1080, 128, seed=777)
        timing("struct1080", l, r, 128, reps=4)
