#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_random.py tests/test_gpu_stages.py -m gpu -q 2>&1 | tail -3
L0=adcensus_amd/lib/skew0/libadcensus_hip.so
for SZ in "" "1242 375"; do
  for rep in 1 2; do
  echo -n "skew0 slack0: "; ADC_HIP_LIB=$L0 ADC_IRV_SLACK=0 timeout 100 python tools/gpu_k8_stats.py $SZ
  echo -n "skew37 slack0: "; ADC_IRV_SLACK=0 timeout 100 python tools/gpu_k8_stats.py $SZ
  echo -n "skew0 slack 16/2: "; ADC_HIP_LIB=$L0 ADC_IRV_SLACK_FMIN=16 ADC_IRV_SLACK_R=2 timeout 100 python tools/gpu_k8_stats.py $SZ
  for CFG in "16 2" "16 0" "32 2" "65 255" "8 2"; do set -- $CFG
    echo -n "skew37 slack $1/$2: "; ADC_IRV_SLACK_FMIN=$1 ADC_IRV_SLACK_R=$2 timeout 100 python tools/gpu_k8_stats.py $SZ
  done
  done
done
ADC_HIP_LIB=adcensus_amd/lib/irvt/libadcensus_hip.so ADC_IRV_SLACK=0 timeout 200 python tools/gpu_irv_timing2.py 8 | cut -c1-330
