#!/bin/bash
# round 6, first call: the GPU test tier on the tree as inherited (+ the reference-caller tests) and the default bench lines.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -16 > $O/r6a_gpu_pytest.log; cat $O/r6a_gpu_pytest.log
timeout 400 python bench.py > $O/r6a_bench_default.json 2> $O/r6a_bench_default.err; echo "default rc=$?"; cut -c1-400 $O/r6a_bench_default.json
timeout 300 python bench.py --workload structured --steps 10 --no-cpu-baseline > $O/r6a_bench_structured.json 2> $O/r6a_bench_structured.err; echo "structured rc=$?"; cut -c1-300 $O/r6a_bench_structured.json
