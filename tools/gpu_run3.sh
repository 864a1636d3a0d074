#!/bin/bash
# quick: parity tests on a subset + timings
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/diag.jsonl
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -k "${PYTEST_K:-s2_ or cone or q_}" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/pytest_gpu.log
timeout 600 python tools/gpu_diag.py timing > gpurun_out/diag_timing.log 2>&1; echo "timing rc=$?"
