#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -40 > gpurun_out/g5_pytest.log; cat gpurun_out/g5_pytest.log
