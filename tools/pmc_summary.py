#!/usr/bin/env python3
"""Summarises the two rocprofv3 --pmc passes of tools/gpu_pmc.sh for the aggregation kernel.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced stream
(MI355X_MICROARCH.md, HBM section) -> doubled here; WRITE_SIZE matched the known byte count (V) exactly."""
import collections
import csv
import json
import sys

wl = sys.argv[1] if len(sys.argv) > 1 else "noise"
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/pmc_%s_%s/pmc_counter_collection.csv" % (wl, C))):
        if r["Counter_Name"] == C and ("k_agg_march" in r["Kernel_Name"] or "k_agg_regring" in r["Kernel_Name"]):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)
    out[C] = {k: sum(v) / len(v) for k, v in agg.items()}
per = {}
for k in out["FETCH_SIZE"]:
    per[k] = {"fetch_bytes_corrected": 2.0 * out["FETCH_SIZE"][k], "write_bytes": out["WRITE_SIZE"].get(k, 0.0)}
    per[k]["total"] = per[k]["fetch_bytes_corrected"] + per[k]["write_bytes"]
per = {k: v for k, v in per.items() if v["write_bytes"] > 1e6}  # drop the variant that exits immediately


def costin(name):  # k_agg_march<VERT, DIVIDE, SMALL, COSTIN, PAIR, VPL>: the fused first pass only writes the volume
    if "k_agg_regring_cost" in name:
        return True
    if "<" not in name:
        return False
    if "k_agg_regring" in name:
        return False
    args = name[name.index("<") + 1:name.rindex(">")].split(",")
    return len(args) >= 4 and args[3].strip() == "true"


regular = {k: v for k, v in per.items() if not costin(k)}
avg = sum(v["total"] for v in regular.values()) / max(1, len(regular))
print(json.dumps({"workload": wl, "per_kernel": per, "traffic_bytes_per_launch_avg": avg,
                  "note": "average over the regular launches (the fused-cost first pass, write-only, is listed but not averaged); FETCH_SIZE x2 (gfx950 calibration), separate --pmc passes, bench.py --steps 2 --inflight 1"}, indent=1))
