#!/bin/bash
# round-2 GPU session: parity gate, then A/B of the aggregation variants on the structured pair + kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/s2_pytest.log; cat gpurun_out/s2_pytest.log
for V in "ADC_AGG_PAIR_FULL=1" "ADC_AGG_PAIR_FULL=0"; do
  echo "== structured $V"
  env $V timeout 300 python bench.py --steps 10 --warmup 3 --workload structured --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; o=json.loads(sys.stdin.read()); print(o["value"], o["stage_ms"], o["roofline"]["avg_launch_ms"])'
done
echo "== noise"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; o=json.loads(sys.stdin.read()); print(o["value"], o["stage_ms"], o["roofline"]["avg_launch_ms"])'
cd /tmp && export TMPDIR=/tmp
for WL in structured; do
  rm -rf "$REPO/gpurun_out/prof_$WL"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_$WL" -o bench -- python "$REPO/bench.py" --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline --workload $WL > "$REPO/gpurun_out/rocprof_$WL.log" 2>&1; echo "rocprof rc=$?"
done
cd "$REPO"; python tools/prof_summary.py $(ls gpurun_out/prof_structured/*/*.db 2>/dev/null | tail -1) 2>&1 | head -40 | tee gpurun_out/s2_kernel_stats_structured.md
