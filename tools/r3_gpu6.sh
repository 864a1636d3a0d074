#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_paper.py tests/test_gpu_api.py -m gpu -x -q --durations=5 2>&1 | tail -25 > gpurun_out/g6_pytest.log; cat gpurun_out/g6_pytest.log
