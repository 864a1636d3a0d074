#!/usr/bin/env python3
"""Summarises the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) of a bench.py run
for the aggregation kernels:  python tools/pmc_summary.py <tag>   with gpurun_out/pmc_<tag>_{FETCH,WRITE}_SIZE/.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced stream
(MI355X_MICROARCH.md, HBM section) -> doubled here; WRITE_SIZE matched the known byte count (V) exactly in round 1."""
import collections
import csv
import glob
import json
import sys

import hashlib
import os

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _n in ("k_aggregate.hip", "k_aggregate_rr.h", "k_aggregate_rr2.h"):  # == bench.py K4_SOURCES / k4_source_hash()
    with open(os.path.join(ROOT, "adcensus_amd", "csrc", _n), "rb") as _f:
        _h.update(_f.read())
rev = _h.hexdigest()[:16]
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    path = glob.glob("gpurun_out/pmc_%s_%s/**/pmc_counter_collection.csv" % (tag, C), recursive=True)[0]
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == C and any(k in r["Kernel_Name"] for k in ("k_agg_march", "k_agg_regring", "k_agg_rr2")):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)
    out[C] = {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}
per = {}
for k in out["FETCH_SIZE"]:
    per[k] = {"fetch_bytes_corrected": 2.0 * out["FETCH_SIZE"][k][0], "write_bytes": out["WRITE_SIZE"].get(k, (0.0, 0))[0],
              "launches_profiled": out["FETCH_SIZE"][k][1]}
    per[k]["total"] = per[k]["fetch_bytes_corrected"] + per[k]["write_bytes"]
per = {k: v for k, v in per.items() if v["write_bytes"] > 1e6}  # drop variants that exit immediately


def costin(name):  # the fused first pass only writes the volume
    if "k_agg_regring_cost" in name or "k_agg_rr2_cost" in name:
        return True
    if "k_agg_regring" in name or "k_agg_rr2" in name or "<" not in name:
        return False
    args = name[name.index("<") + 1:name.rindex(">")].split(",")
    return len(args) >= 4 and args[3].strip() == "true"


regular = {k: v for k, v in per.items() if not costin(k)}
n = sum(v["launches_profiled"] for v in regular.values())
avg = sum(v["total"] * v["launches_profiled"] for v in regular.values()) / max(1, n)
print(json.dumps({"tag": tag, "k4_src_sha16": rev, "per_kernel": per, "traffic_bytes_per_launch_avg": avg,
                  "note": "launch-weighted average over the regular launches (the fused-cost first pass, write-only, is listed but not "
                          "averaged); FETCH_SIZE x2 (gfx950 calibration), separate --pmc passes, bench.py --steps 2 --warmup 1"}, indent=1))
