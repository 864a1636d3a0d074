#!/bin/bash
# 60-second look at the two round-3 experiment patches on a GPU, through a variant library built with both applied
# (adcensus_amd/lib/exp/libadcensus_hip.so, ADC_HIP_LIB): a few stage cases with each form switched on, then one bench line.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
export ADC_HIP_LIB=$GRAFT_REPO_ROOT/adcensus_amd/lib/exp/libadcensus_hip.so
K="s2_96x64_d32 or cone_crop_d40 or q_257x131_d64 or noise_160x90_d128"
ADC_MEDIAN_JACOBI=12 timeout 22 python -m pytest tests/test_gpu_stages.py -m gpu -x -q -k "$K" 2>&1 | tail -4 > $O/quick_medj.log; echo "median:"; cat $O/quick_medj.log
ADC_INTERP_REFILL=4096 timeout 22 python -m pytest tests/test_gpu_stages.py -m gpu -x -q -k "$K" 2>&1 | tail -4 > $O/quick_refill.log; echo "refill:"; cat $O/quick_refill.log
for CFG in "0 0" "12 0" "0 4096"; do
  set -- $CFG
  ADC_MEDIAN_JACOBI=$1 ADC_INTERP_REFILL=$2 timeout 12 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/quick_$1_$2.json 2> $O/quick_err.txt
  python -c "import json; d=json.load(open('$O/quick_$1_$2.json')); print('jacobi=$1 refill=$2', round(d['value'],1), 'refine ms', d['stage_ms']['refine'], d['async_fallbacks'], d['farm_check']['ok'], d['farm_check']['committed_1gpu_mismatches'][:3])" 2>&1 | tail -1
done
