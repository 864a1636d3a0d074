#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for CFG in "1 0" "2 1" "3 1" "3 0"; do
  set -- $CFG; F=$1; SH=$2
  for WL in noise structured; do
    ADC_SHARED_HEAVY=$SH timeout 600 python bench.py --inflight $F --workload $WL --steps 24 --warmup 4 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c '
import sys,json
o=json.loads(sys.stdin.read()); r=o["roofline"]
print("F='$F' shared='$SH' '$WL'", o["value"], "ms/step", o["ms_per_step"], "farm", o["farm_check"]["ok"], "agg_ms", o["stage_ms"].get("aggregate"), "launch_ms", r["avg_launch_ms"], "frac", r["frac"], "hbm_frac", r["hbm_frac"])'
  done
done
