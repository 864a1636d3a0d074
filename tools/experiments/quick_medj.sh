#!/bin/bash
# short look at the blocked-Jacobi median through the variant library (see quick_both.sh)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
export ADC_HIP_LIB=$GRAFT_REPO_ROOT/adcensus_amd/lib/exp/libadcensus_hip.so
K="s2_96x64_d32 or cone_crop_d40 or q_257x131_d64 or noise_160x90_d128 or q_9x20_d8 or q_30x7_d8"
ADC_MEDIAN_JACOBI=12 timeout 15 python -m pytest tests/test_gpu_stages.py -m gpu -x -q -k "$K" 2>&1 | tail -12 | cut -c1-250 > $O/quick_medj.log; cat $O/quick_medj.log
for J in 12 0; do
  ADC_MEDIAN_JACOBI=$J timeout 10 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/quick_${J}_0.json 2> $O/quick_err.txt
  python -c "import json; d=json.load(open('$O/quick_${J}_0.json')); print('jacobi=$J', round(d['value'],1), 'refine ms', d['stage_ms']['refine'], d['async_fallbacks'], d['farm_check']['ok'], d['farm_check']['committed_1gpu_mismatches'][:3])" 2>&1 | tail -1
done
