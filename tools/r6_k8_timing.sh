#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export PYTHONUNBUFFERED=1
L=adcensus_amd/lib/irvt/libadcensus_hip.so
ADC_HIP_LIB=$L ADC_IRV_SLACK=0 timeout 200 python tools/gpu_irv_timing2.py 8
