#!/bin/bash
# round 4, validation of the FINAL tree beyond the GPU test tier (which tools/r4_final.sh runs on the same tree): the cross-check of
# the speculative forms against the plain ones on many different pairs, and the mixed stress of the asynchronous pipeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 400 python tools/gpu_speculation_check.py 16 > $O/r4_speculation_check.txt 2>&1; echo "speculation check rc=$?"; cat $O/r4_speculation_check.txt
timeout 300 python tools/gpu_stress_mixed.py 12 > $O/r4_stress_mixed.txt 2>&1; echo "stress rc=$?"; tail -4 $O/r4_stress_mixed.txt
