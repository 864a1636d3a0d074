#!/bin/bash
# round 6, call 3: the whole GPU tier on the tree with the voting chain's slack budgets, then the structured workload (1080p and KITTI size)
# with the budgets on / off, interleaved on the same box.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -14 > $O/r6c_gpu_pytest.log; cat $O/r6c_gpu_pytest.log
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; python - "$O/sw_$tag.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.strip().startswith("{")][-1])
print("      voting:", d.get("async_fallbacks", {}).get("voting_chain_budget"), d.get("async_fallbacks", {}).get("voting_continuations"))
PY
}
for rep in 1 2; do
  ARGS="--steps 10 $B --workload structured"
  run k8_s1080_slack1_$rep ADC_IRV_SLACK=1
  run k8_s1080_slack0_$rep ADC_IRV_SLACK=0
  ARGS="--width 1242 --height 375 --steps 30 $B --workload structured"
  run k8_skitti_slack1_$rep ADC_IRV_SLACK=1
  run k8_skitti_slack0_$rep ADC_IRV_SLACK=0
  ARGS="--steps 20 $B --workload noise"
  run k8_n1080_slack1_$rep ADC_IRV_SLACK=1
  run k8_n1080_slack0_$rep ADC_IRV_SLACK=0
done
