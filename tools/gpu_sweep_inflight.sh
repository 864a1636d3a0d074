#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for F in 1 2 3 4; do
  for WL in noise structured; do
    timeout 600 python bench.py --inflight $F --workload $WL --steps 24 --warmup 4 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c '
import sys,json
o=json.loads(sys.stdin.read()); r=o["roofline"]
print("F='$F' '$WL'", o["value"], "ms/step", o["ms_per_step"], "farm", o["farm_check"]["ok"], "agg_ms", o["stage_ms"].get("aggregate"), "launch_ms", r["avg_launch_ms"], "frac", r["frac"], "hbm_frac", r["hbm_frac"])'
  done
done
