#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for f in 1 2 3; do
  timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --inflight $f 2>&1 | tail -1 | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); print('noise inflight $f: %.1f pairs/s, k4 %.4f ms' % (o['value'], o['roofline']['avg_launch_ms']))"
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --inflight $f --workload structured 2>&1 | tail -1 | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); print('structured inflight $f: %.1f pairs/s' % (o['value']))"
done
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json; cut -c1-300 gpurun_out/bench_default.json
