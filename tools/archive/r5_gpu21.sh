#!/bin/bash
O=gpurun_out/r5_21; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fused_tail.py -x -q > $O/pytest_fused.log 2>&1; echo "fused rc=$?"; tail -15 $O/pytest_fused.log
