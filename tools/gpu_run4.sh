#!/bin/bash
# bench A/B over in-flight counts and overlap policies
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for excl in 1 0; do for nf in 1 2 3 4; do
  echo "== ADC_HEAVY_EXCLUSIVE=$excl inflight=$nf noise"
  ADC_HEAVY_EXCLUSIVE=$excl timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --inflight $nf 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], 'pairs/s', 'agg_ms', o['roofline']['avg_launch_ms'], 'frac', o['roofline']['frac'], o['stage_ms'])"
done; done
for excl in 1 0; do for nf in 2 3 4; do
  echo "== ADC_HEAVY_EXCLUSIVE=$excl inflight=$nf structured"
  ADC_HEAVY_EXCLUSIVE=$excl timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --inflight $nf --workload structured 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], 'pairs/s', 'agg_ms', o['roofline']['avg_launch_ms'], 'frac', o['roofline']['frac'], o['stage_ms'])"
done; done
