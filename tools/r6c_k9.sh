#!/bin/bash
# round 6, second session: K9 ray walk on validity tiles -- parity (stage tests, random geometries, full size) and same-box A/B against
# the library of the commit before (adcensus_amd/lib/r6base), interleaved; the walk on the padded code map.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -8 > $O/r6c_gpu_pytest.log; cat $O/r6c_gpu_pytest.log
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
BASE=ADC_HIP_LIB=$GRAFT_REPO_ROOT/adcensus_amd/lib/r6base/libadcensus_hip.so
for rep in 1 2; do
  ARGS="--steps 20 $B --workload noise"
  run k9_noise_base_$rep $BASE
  run k9_noise_new_$rep X=1
  ARGS="--steps 10 $B --workload structured"
  run k9_struct_base_$rep $BASE
  run k9_struct_new_$rep X=1
  ARGS="--width 1242 --height 375 --steps 40 $B --workload noise"
  run k9_kitti_noise_base_$rep $BASE
  run k9_kitti_noise_new_$rep X=1
done
