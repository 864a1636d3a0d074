#!/bin/bash
# round 6, second session: slim margin kernels of the voting chain -- parity (whole chain slim / from the middle) and timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "slim or budgets or voting" 2>&1 | tail -4
ADC_IRV_SLIM_FROM=2 timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -2
ADC_IRV_SLIM_FROM=0 timeout 900 python -m pytest tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -2
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
for rep in 1 2 3; do
  ARGS="--steps 10 $B --workload structured"
  run slim_struct_off_$rep ADC_IRV_SLIM=0
  run slim_struct_on_$rep X=1
done
ARGS="--width 1242 --height 375 --steps 20 $B --workload structured"
run slim_kitti_struct_off ADC_IRV_SLIM=0
run slim_kitti_struct_on X=1
ARGS="--steps 20 $B --workload noise"
run slim_noise_off ADC_IRV_SLIM=0
run slim_noise_on X=1
