#!/bin/bash
# round 4, validation of the FINAL tree (one gpurun call): the whole GPU test tier, the cross-check of the speculative forms against
# the plain ones on many different pairs, the mixed stress of the asynchronous pipeline, one default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -14 > $O/r4_gpu_pytest_final.log; cat $O/r4_gpu_pytest_final.log
grep -q " passed" $O/r4_gpu_pytest_final.log && ! grep -q "failed\|error" $O/r4_gpu_pytest_final.log || { echo "TESTS NOT GREEN -- stopping"; exit 1; }
timeout 400 python tools/gpu_speculation_check.py 16 > $O/r4_speculation_check.txt 2>&1; echo "speculation check rc=$?"; cat $O/r4_speculation_check.txt
timeout 300 python tools/gpu_stress_mixed.py 12 > $O/r4_stress_mixed.txt 2>&1; echo "stress rc=$?"; tail -4 $O/r4_stress_mixed.txt
timeout 200 python bench.py --no-cpu-baseline > $O/r4_bench_final_tree.json 2> $O/r4_bench_final_tree.err; echo "bench rc=$?"; cut -c1-260 $O/r4_bench_final_tree.json
