#!/bin/bash
# Next-round experiment (NOT run yet): K9 with rays as the unit of work (tools/itp_ray_stats.py).  From the repo root:
#   git apply tools/experiments/interp_ray_refill.patch && make -C adcensus_amd/csrc
#   gpurun --timeout 600 -- 'bash tools/experiments/interp_ray_refill.sh'
# Stops at the first failing step.  ADC_INTERP_REFILL = rays per wave range (>= 64; 4096 = 256 targets per range).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
ADC_INTERP_REFILL=4096 timeout 300 python -m pytest tests/test_gpu_stages.py -m gpu -x -q 2>&1 | tail -5 | tee $O/refill_pytest.log
grep -q " passed" $O/refill_pytest.log && ! grep -q "failed\|error" $O/refill_pytest.log || { echo "NOT GREEN -- stopping"; exit 1; }
ADC_INTERP_REFILL=4096 timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -5 | tee -a $O/refill_pytest.log
tail -1 $O/refill_pytest.log | grep -q " passed" || { echo "NOT GREEN -- stopping"; exit 1; }
for rep in 1 2; do
 for R in 0 1024 4096 16384; do
    ADC_INTERP_REFILL=$R timeout 120 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/refill_$R_$rep.json 2> $O/refill_err.txt || { tail -3 $O/refill_err.txt; exit 1; }
    python -c "import json; d=json.load(open('$O/refill_$R_$rep.json')); print('refill=$R', round(d['value'],1), 'refine ms', d['stage_ms']['refine'])"
 done
done
