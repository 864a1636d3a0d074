#!/bin/bash
# Next-round experiment: second form of the tiled median (ring-padded maps: every tile interior; tiles skipped while their 3x3
# neighbourhood is quiet).  NOT run on a GPU yet; its geometry and per-pixel functions are emulated in tests/test_emul.py.
#   git apply tools/experiments/median_tiles_v2.patch && make -C adcensus_amd/csrc
#   gpurun --timeout 420 -- 'bash tools/experiments/median_tiles_v2.sh'
# Stops at the first failing step.  ADC_MEDIAN_TILES = kernels in the chain (budget; a noise seed needs 16 kernels of 8 rounds),
# ADC_MEDIAN_TILE = 32 | 64.  First the median stage alone on a few cases, then the parity tiers, then timing (per-kernel times:
# add a rocprofv3 --kernel-trace --stats run of one bench line).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
for TS in 64 32; do
  ADC_MEDIAN_TILES=24 ADC_MEDIAN_TILE=$TS timeout 200 python -m pytest tests/test_gpu_stages.py -m gpu -x -q 2>&1 | tail -5 | tee $O/medp_pytest_$TS.log
  grep -q " passed" $O/medp_pytest_$TS.log && ! grep -q "failed\|error" $O/medp_pytest_$TS.log || { echo "NOT GREEN ($TS) -- stopping"; exit 1; }
done
for rep in 1 2; do
 for CFG in "0 64" "24 64" "24 32"; do
  set -- $CFG
  for WL in noise structured; do
    ADC_MEDIAN_TILES=$1 ADC_MEDIAN_TILE=$2 timeout 60 python bench.py --workload $WL --steps 12 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/medp_${WL}_$1_$2_$rep.json 2> $O/medp_err.txt || { tail -3 $O/medp_err.txt; exit 1; }
    python -c "import json; d=json.load(open('$O/medp_${WL}_$1_$2_$rep.json')); print('tiles=$1 S=$2', '$WL', round(d['value'],1), 'refine ms', d['stage_ms']['refine'], d['async_fallbacks']['median_handoff'], d['farm_check']['ok'])"
  done
 done
done
