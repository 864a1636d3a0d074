#!/bin/bash
# round 5, validation of the final tree beyond tools/r5_final.sh: random geometries / options against the CPU oracle (voting stage in
# isolation + whole Match), and the cross-check of the speculative forms against the plain ones on 48 different pairs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_random.py -x -q > $O/r5_gpu_pytest_random.log 2>&1; echo "random rc=$?"; tail -4 $O/r5_gpu_pytest_random.log
timeout 400 python tools/gpu_speculation_check.py 16 > $O/r5_speculation_check.txt 2>&1; echo "speculation check rc=$?"; tail -6 $O/r5_speculation_check.txt
