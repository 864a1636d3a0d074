#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
L=adcensus_amd/lib
for S in "1080 --steps 10" "kitti --width 1242 --height 375 --steps 30"; do
  set -- $S; T=$1; shift
  ARGS="$* $B --workload structured"
  for rep in 1 2; do
  run k8g_${T}_slack1_$rep X=1
  run k8g_${T}_skew37_$rep ADC_HIP_LIB=$L/skew37/libadcensus_hip.so
  run k8g_${T}_skew13_$rep ADC_HIP_LIB=$L/skew13/libadcensus_hip.so
  run k8g_${T}_wpb16_$rep ADC_IRV_WPB=16
  run k8g_${T}_wpb4_$rep ADC_IRV_WPB=4
  run k8g_${T}_grid512_$rep ADC_IRV_GRID=512
  done
done
