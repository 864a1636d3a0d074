#!/usr/bin/env python3
"""Per-kernel SQ counter table from separate rocprofv3 --pmc passes (kernel-trace only):
    python tools/pmc_sq_summary.py gpurun_out/<prefix>_   ->  reads <prefix>_1, <prefix>_2, ...
Prints, per kernel: dispatches, average duration (under the counter pass), shares of the wave cycles and instructions per wave."""
import collections
import csv
import glob
import sys

prefix = sys.argv[1]
only = sys.argv[2:]
tot = collections.defaultdict(collections.Counter)
ndisp = collections.Counter()
dur = collections.Counter()
first = True
for d in sorted(glob.glob(prefix + "[0-9]*")):
    if d.endswith(".log"):
        continue
    for f in glob.glob(d + "/**/pmc_counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if only and not any(o in k for o in only):
                continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if first and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                ndisp[k] += 1
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    first = False
print("| kernel | n | avg us | wait_any | wait_inst_any | valu | salu | lds | vmem | VALU/wave | SALU/wave | VMEM_RD/wave | waves |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for k in sorted(tot, key=lambda k: -dur[k]):
    t = tot[k]
    wc = max(t["SQ_WAVE_CYCLES"], 1.0)
    waves = max(t.get("SQ_WAVES", 0.0), 1.0)
    print("| `%s` | %d | %.1f | %.2f | %.2f | %.3f | %.3f | %.3f | %.3f | %.0f | %.0f | %.0f | %.0f |" % (
        k[:70], ndisp[k], dur[k] / max(1, ndisp[k]), t["SQ_WAIT_ANY"] / wc, t["SQ_WAIT_INST_ANY"] / wc, t["SQ_ACTIVE_INST_VALU"] / wc,
        t["SQ_ACTIVE_INST_SCA"] / wc, t["SQ_ACTIVE_INST_LDS"] / wc, t["SQ_ACTIVE_INST_VMEM"] / wc,
        t["SQ_INSTS_VALU"] / waves, t["SQ_INSTS_SALU"] / waves, t["SQ_INSTS_VMEM_RD"] / waves, t.get("SQ_WAVES", 0.0) / max(1, ndisp[k])))
