#!/bin/bash
# Next-round experiment (NOT run yet): the in-place median as a temporally blocked chaotic iteration (DESIGN 4.4,
# tools/median_rounds.py).  From the repo root:
#   git apply tools/experiments/median_blocked_jacobi.patch && make -C adcensus_amd/csrc
#   gpurun --timeout 600 -- 'bash tools/experiments/median_blocked_jacobi.sh'
# Stops at the first failing step.  ADC_MEDIAN_JACOBI = number of kernels in the chain (8 rounds each; the 1080p noise pair needs
# 8-9, one noise seed 16: the budget below is 24); a chain that has not converged takes the existing fallback in adc_wait, so a wrong result can only
# come from the kernel itself.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
ADC_MEDIAN_JACOBI=24 timeout 300 python -m pytest tests/test_gpu_stages.py -m gpu -x -q 2>&1 | tail -5 | tee $O/medj_pytest.log
grep -q " passed" $O/medj_pytest.log && ! grep -q "failed\|error" $O/medj_pytest.log || { echo "NOT GREEN -- stopping"; exit 1; }
ADC_MEDIAN_JACOBI=24 timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -5 | tee -a $O/medj_pytest.log
tail -1 $O/medj_pytest.log | grep -q " passed" || { echo "NOT GREEN -- stopping"; exit 1; }
for rep in 1 2; do
 for J in 0 24 16; do
  for WL in noise structured; do
    ADC_MEDIAN_JACOBI=$J timeout 120 python bench.py --workload $WL --steps 12 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/medj_${WL}_$J_$rep.json 2> $O/medj_err.txt || { tail -3 $O/medj_err.txt; exit 1; }
    python -c "import json; d=json.load(open('$O/medj_${WL}_$J_$rep.json')); print('jacobi=$J', '$WL', round(d['value'],1), 'refine ms', d['stage_ms']['refine'], d['async_fallbacks'])"
  done
 done
done
