#!/bin/bash
# round 6, second session: K9 on the code map, finished rays masked out -- steps of the first trip / of the following trips, same box, interleaved
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out; L=$GRAFT_REPO_ROOT/adcensus_amd/lib
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
for rep in 1 2; do
  ARGS="--steps 20 $B --workload noise"
  run k9c_noise_base_$rep ADC_HIP_LIB=$L/r6base/libadcensus_hip.so
  for v in k9_4_8 k9_4_4 k9_3_6 k9_2_8 k9_2_4 k9_3_3 k9_6_8; do run k9c_noise_${v}_$rep ADC_HIP_LIB=$L/$v/libadcensus_hip.so; done
done
ARGS="--width 1242 --height 375 --steps 40 $B --workload noise"
run k9c_kitti_noise_base ADC_HIP_LIB=$L/r6base/libadcensus_hip.so
for v in k9_4_8 k9_4_4 k9_3_6 k9_2_8 k9_2_4; do run k9c_kitti_noise_${v} ADC_HIP_LIB=$L/$v/libadcensus_hip.so; done
ARGS="--steps 10 $B --workload structured"
run k9c_struct_base ADC_HIP_LIB=$L/r6base/libadcensus_hip.so
for v in k9_4_8 k9_4_4 k9_2_4; do run k9c_struct_${v} ADC_HIP_LIB=$L/$v/libadcensus_hip.so; done
