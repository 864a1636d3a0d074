#!/bin/bash
# round 6, second session: k_so_c1 column passes with coalesced loads, voting budget of empty lists -- parity + timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out; L=$GRAFT_REPO_ROOT/adcensus_amd/lib
timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_random.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -4
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
for rep in 1 2 3; do
  ARGS="--steps 20 $B --workload noise"
  run misc_noise_prev_$rep ADC_HIP_LIB=$L/r6prev2/libadcensus_hip.so
  run misc_noise_new_$rep X=1
done
ARGS="--steps 10 $B --workload structured"
run misc_struct_prev ADC_HIP_LIB=$L/r6prev2/libadcensus_hip.so
run misc_struct_new X=1
ARGS="--width 1242 --height 375 --steps 40 $B --workload noise"
run misc_kitti_noise_prev ADC_HIP_LIB=$L/r6prev2/libadcensus_hip.so
run misc_kitti_noise_new X=1
