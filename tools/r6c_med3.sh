#!/bin/bash
# round 6, second session: median segments with staggered chain links -- parity (median variants) and the head start per link
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out; L=$GRAFT_REPO_ROOT/adcensus_amd/lib
timeout 1500 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "median" 2>&1 | tail -4 > $O/r6c_med3_pytest.log; cat $O/r6c_med3_pytest.log
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
for rep in 1 2; do
  ARGS="--steps 20 $B --workload noise"
  run med3_noise_new_$rep X=1
  for v in med_l0 med_l32 med_l96; do run med3_noise_${v}_$rep ADC_HIP_LIB=$L/$v/libadcensus_hip.so; done
  run med3_noise_seg4_$rep ADC_MEDIAN_SEG=4
  run med3_noise_seg6_$rep ADC_MEDIAN_SEG=6
done
ARGS="--width 1242 --height 375 --steps 40 $B --workload noise"
run med3_kitti_noise_new X=1
run med3_kitti_noise_l96 ADC_HIP_LIB=$L/med_l96/libadcensus_hip.so
run med3_kitti_noise_seg8 ADC_MEDIAN_SEG=8
run med3_kitti_noise_seg3 ADC_MEDIAN_SEG=3
