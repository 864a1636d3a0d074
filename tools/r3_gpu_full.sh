#!/bin/bash
# full GPU test tier on the current tree; the tail of the log is what profiles/r3_gpu_pytest.log holds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 > gpurun_out/r3_gpu_pytest.log; cat gpurun_out/r3_gpu_pytest.log
