#!/bin/bash
# round 6, second session: K11 median with speculative column segments -- parity (segment variants, band variants, stage tests) and
# same-box A/B against the library of the commit before; segment counts / warm-up.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out; L=$GRAFT_REPO_ROOT/adcensus_amd/lib
timeout 1500 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "median" 2>&1 | tail -8 > $O/r6c_med_pytest.log; cat $O/r6c_med_pytest.log
timeout 900 python -m pytest tests/test_gpu_stages.py -m gpu -x -q 2>&1 | tail -4
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
for rep in 1 2; do
  ARGS="--steps 20 $B --workload noise"
  run med_noise_base_$rep ADC_HIP_LIB=$L/r6base/libadcensus_hip.so
  run med_noise_new_$rep X=1
  run med_noise_seg1_$rep ADC_MEDIAN_SEG=1
  run med_noise_seg4_$rep ADC_MEDIAN_SEG=4
  run med_noise_seg6_$rep ADC_MEDIAN_SEG=6
  run med_noise_seg8w64_$rep ADC_MEDIAN_SEG=8 ADC_MEDIAN_WARM=64
  run med_noise_seg8w192_$rep ADC_MEDIAN_SEG=8 ADC_MEDIAN_WARM=192
done
ARGS="--steps 10 $B --workload structured"
run med_struct_base ADC_HIP_LIB=$L/r6base/libadcensus_hip.so
run med_struct_new X=1
ARGS="--width 1242 --height 375 --steps 40 $B --workload noise"
run med_kitti_noise_base ADC_HIP_LIB=$L/r6base/libadcensus_hip.so
run med_kitti_noise_new X=1
run med_kitti_noise_seg3 ADC_MEDIAN_SEG=3
run med_kitti_noise_seg8 ADC_MEDIAN_SEG=8
ARGS="--width 1242 --height 375 --steps 20 $B --workload structured"
run med_kitti_struct_base ADC_HIP_LIB=$L/r6base/libadcensus_hip.so
run med_kitti_struct_new X=1
