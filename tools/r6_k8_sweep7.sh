#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
timeout 900 python -m pytest tests/test_gpu_random.py tests/test_gpu_stages.py -m gpu -q 2>&1 | tail -3
L=adcensus_amd/lib
for S in "1080 --steps 10" "kitti --width 1242 --height 375 --steps 30"; do
  set -- $S; T=$1; shift
  ARGS="$* $B --workload structured"
  for rep in 1 2 3; do
  run k8e_${T}_r5lib_$rep ADC_HIP_LIB=$L/r5/libadcensus_hip.so
  run k8e_${T}_slack0_$rep ADC_IRV_SLACK=0
  run k8e_${T}_slack1_$rep ADC_IRV_SLACK=1
  done
done
ARGS="--steps 20 $B --workload noise"
for rep in 1 2; do run k8e_n1080_r5lib_$rep ADC_HIP_LIB=$L/r5/libadcensus_hip.so; run k8e_n1080_slack1_$rep X=1; done
