#!/bin/bash
# round 6, second session: median with a narrower first segment -- parity + timing (ADC_MEDIAN_SHIFT=0: equal widths)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "median" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_stages.py -m gpu -x -q 2>&1 | tail -2
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
for rep in 1 2 3; do
  ARGS="--steps 20 $B --workload noise"
  run med4_noise_equal_$rep ADC_MEDIAN_SHIFT=0 ADC_MEDIAN_SEG=8
  run med4_noise_new_$rep X=1
done
ARGS="--width 1242 --height 375 --steps 40 $B --workload noise"
run med4_kitti_equal ADC_MEDIAN_SHIFT=0
run med4_kitti_new X=1
python tools/gpu_median_probe2.py 2>&1 | grep "1080 " | head -4
