#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
timeout 900 python -m pytest tests/test_gpu_random.py tests/test_gpu_stages.py -m gpu -q 2>&1 | tail -3
LX=adcensus_amd/lib/rcexact/libadcensus_hip.so
for S in "1080 --steps 10" "kitti --width 1242 --height 375 --steps 30"; do
  set -- $S; T=$1; shift
  ARGS="$* $B --workload structured"
  for rep in 1 2; do
  run k8c_${T}_slack0_$rep ADC_IRV_SLACK=0
  run k8c_${T}_f16r2_$rep ADC_IRV_SLACK_FMIN=16 ADC_IRV_SLACK_R=2
  run k8c_${T}_f16r0_$rep ADC_IRV_SLACK_FMIN=16 ADC_IRV_SLACK_R=0
  run k8c_${T}_f1r0_$rep ADC_IRV_SLACK_FMIN=1 ADC_IRV_SLACK_R=0
  run k8c_${T}_f1r2_$rep ADC_IRV_SLACK_FMIN=1 ADC_IRV_SLACK_R=2
  run k8c_${T}_f32r2_$rep ADC_IRV_SLACK_FMIN=32 ADC_IRV_SLACK_R=2
  run k8c_${T}_exact_f16r2_$rep ADC_HIP_LIB=$LX ADC_IRV_SLACK_FMIN=16 ADC_IRV_SLACK_R=2
  done
done
