#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
export ADC_HIP_LIB=$GRAFT_REPO_ROOT/adcensus_amd/lib/exp/libadcensus_hip.so
for J in 4 8 16 24; do
  ADC_MEDIAN_JACOBI=$J timeout 6 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/quick_${J}_0.json 2> $O/quick_err.txt
  python -c "import json; d=json.load(open('$O/quick_${J}_0.json')); print('jacobi=$J', round(d['value'],1), 'refine ms', d['stage_ms']['refine'], d['async_fallbacks']['median_handoff'], d['farm_check']['ok'])" 2>&1 | tail -1
done
