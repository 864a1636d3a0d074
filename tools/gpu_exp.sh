#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export PYTHONUNBUFFERED=1
for v in 1 2 3 4 5; do
  echo "== ADC_AGG_VSEG=$v"
  ADC_AGG_VSEG=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --inflight 1 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], 'pairs/s', 'agg_ms', o['roofline']['avg_launch_ms'], 'frac', o['roofline']['frac'], o['stage_ms'])"
done
for hseg in 2 4; do
  echo "== ADC_AGG_HSEG=$hseg"
  ADC_AGG_HSEG=$hseg timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --inflight 1 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], 'pairs/s', 'agg_ms', o['roofline']['avg_launch_ms'], 'frac', o['roofline']['frac'], o['stage_ms'])"
done
