#!/bin/bash
# same-box A/B at 1080p: scanline kernels before (compiler-allocated prefetch slots) / after (pinned slots)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
for rep in 1 2 3; do
 for V in old new; do
  for WL in noise structured; do
    if [ $V = old ]; then export ADC_HIP_LIB=$GRAFT_REPO_ROOT/adcensus_amd/lib/so_old/libadcensus_hip.so; else unset ADC_HIP_LIB; fi
    timeout 150 python bench.py --workload $WL --steps 12 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/k5c_${WL}_${V}_$rep.json 2> $O/k5c_err.txt || { echo "bench failed"; tail -3 $O/k5c_err.txt; exit 1; }
    python - <<PY
import json
d=json.load(open("$O/k5c_${WL}_${V}_$rep.json"))
print("$V", "$WL", "pairs/s", round(d["value"],1), "scanline ms", d["stage_ms"]["scanline"], "agg", d["stage_ms"]["aggregate"], "wta", d["stage_ms"]["wta"])
PY
  done
 done
done
