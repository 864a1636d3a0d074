#!/usr/bin/env python3
"""One line per bench.py JSON line read from the files given: the numbers the same-box sweeps compare."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(path) if l.strip().startswith("{")][-1])
    except Exception as exc:  # noqa: BLE001
        print("%-44s  unreadable (%s)" % (path.split("/")[-1], exc))
        continue
    st, rf = d.get("stage_ms") or {}, d.get("roofline") or {}
    ok = (d.get("farm_check") or {}).get("ok")
    print("%-44s %8.2f pairs/s  %6.3f ms  agg %.3f so %.3f wta %.3f refine %.3f | K4 launch %.4f ms frac %.3f x%s | check %s refmis %s" % (
        path.split("/")[-1], d["value"], d["ms_per_step"], st.get("aggregate", 0), st.get("scanline", 0), st.get("wta", 0), st.get("refine", 0),
        rf.get("avg_launch_ms", 0), rf.get("frac", 0), rf.get("regular_launches"), ok, len((d.get("farm_check") or {}).get("reference_mismatches", []))))
