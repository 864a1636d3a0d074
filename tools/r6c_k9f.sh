#!/bin/bash
# round 6, second session: K9 with K targets per lane and iteration (independent round-trip chains overlap) -- parity + same-box A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out; L=$GRAFT_REPO_ROOT/adcensus_amd/lib
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -2
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
for rep in 1 2; do
  ARGS="--steps 20 $B --workload noise"
  run k9f_noise_k1_$rep ADC_HIP_LIB=$L/k9k1/libadcensus_hip.so
  run k9f_noise_k2_$rep X=1
  run k9f_noise_k3_$rep ADC_HIP_LIB=$L/k9k3/libadcensus_hip.so
  run k9f_noise_k4_$rep ADC_HIP_LIB=$L/k9k4/libadcensus_hip.so
done
ARGS="--width 1242 --height 375 --steps 40 $B --workload noise"
run k9f_kitti_noise_k1 ADC_HIP_LIB=$L/k9k1/libadcensus_hip.so
run k9f_kitti_noise_k2 X=1
run k9f_kitti_noise_k4 ADC_HIP_LIB=$L/k9k4/libadcensus_hip.so
ARGS="--steps 10 $B --workload structured"
run k9f_struct_k1 ADC_HIP_LIB=$L/k9k1/libadcensus_hip.so
run k9f_struct_k2 X=1
