#!/bin/bash
# round 6, second session: median with private chains for the bands 1 .. spec -- parity (median variants) and timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out; L=$GRAFT_REPO_ROOT/adcensus_amd/lib
timeout 1500 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "median" 2>&1 | tail -8 > $O/r6c_med2_pytest.log; cat $O/r6c_med2_pytest.log
timeout 900 python -m pytest tests/test_gpu_stages.py -m gpu -x -q 2>&1 | tail -3
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
for rep in 1 2; do
  ARGS="--steps 20 $B --workload noise"
  run med2_noise_base_$rep ADC_HIP_LIB=$L/r6base/libadcensus_hip.so
  run med2_noise_new_$rep X=1
  run med2_noise_seg4_$rep ADC_MEDIAN_SEG=4
  run med2_noise_seg6_$rep ADC_MEDIAN_SEG=6
  run med2_noise_seg8w64_$rep ADC_MEDIAN_SEG=8 ADC_MEDIAN_WARM=64
  run med2_noise_spec1_$rep ADC_MEDIAN_SPEC=1
done
ARGS="--width 1242 --height 375 --steps 40 $B --workload noise"
run med2_kitti_noise_base ADC_HIP_LIB=$L/r6base/libadcensus_hip.so
run med2_kitti_noise_new X=1
run med2_kitti_noise_seg8 ADC_MEDIAN_SEG=8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/$O/prof_med2
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_med2 -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --workload noise > $R/$O/rocprof_med2.log 2>&1
cd $R; python tools/prof_summary.py $(ls $O/prof_med2/*.db $O/prof_med2/*/*.db 2>/dev/null | tail -1) 2>&1 | grep -i "median\|interp\|itp" | cut -c1-120
