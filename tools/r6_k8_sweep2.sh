#!/bin/bash
# round 6: voting chain with slack budgets on the exact region count -- band height (variant libraries), workgroup shape, rows per trip;
# structured 1080p + KITTI size, same box, default first and last.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
timeout 600 python -m pytest tests/test_gpu_random.py tests/test_gpu_stages.py tests/test_gpu_api.py tests/test_gpu_faults.py -m gpu -q 2>&1 | tail -3
L=adcensus_amd/lib
for S in "1080 --steps 10" "kitti --width 1242 --height 375 --steps 30"; do
  set -- $S; T=$1; shift
  ARGS="$* $B --workload structured"
  run k8b_${T}_default X=1
  run k8b_${T}_slack0 ADC_IRV_SLACK=0
  run k8b_${T}_band4 ADC_HIP_LIB=$L/band4/libadcensus_hip.so
  run k8b_${T}_band16 ADC_HIP_LIB=$L/band16/libadcensus_hip.so
  run k8b_${T}_band32 ADC_HIP_LIB=$L/band32/libadcensus_hip.so
  run k8b_${T}_wpb4 ADC_IRV_WPB=4
  run k8b_${T}_wpb16 ADC_IRV_WPB=16
  run k8b_${T}_grid512 ADC_IRV_GRID=512
  run k8b_${T}_grid2048 ADC_IRV_GRID=2048
  run k8b_${T}_grid2048_wpb4 ADC_IRV_GRID=2048 ADC_IRV_WPB=4
  run k8b_${T}_default_again X=1
done
