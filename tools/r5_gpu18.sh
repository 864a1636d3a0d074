#!/bin/bash
O=gpurun_out/r5_18; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_random.py -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
timeout 300 python tools/gpu_stress_mixed.py 4 2>&1 | tail -2
