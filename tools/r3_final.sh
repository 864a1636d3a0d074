#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16 > gpurun_out/r3_gpu_pytest.log; cat gpurun_out/r3_gpu_pytest.log
timeout 300 python tools/gpu_median_probe.py > gpurun_out/r3_median_probe.txt 2>&1; cat gpurun_out/r3_median_probe.txt
bash tools/gpu_profile_r3.sh
