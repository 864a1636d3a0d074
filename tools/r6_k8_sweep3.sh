#!/bin/bash
# round 6: hybrid slack filter of the voting chain -- slack_r (entries kept as "maybes") x slack_fmin (hit entries per wave from which the wave
# filters in phase 1); structured 1080p + KITTI size; same box.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_random.py tests/test_gpu_stages.py tests/test_gpu_api.py tests/test_gpu_faults.py -m gpu -q 2>&1 | tail -3
for SZ in "" "1242 375"; do
  ADC_IRV_SLACK=0 timeout 100 python tools/gpu_k8_stats.py $SZ
  for FM in 1 8 16 32 65; do
    for R in 0 2 4 8 255; do
      echo -n "fmin $FM r $R: "; ADC_IRV_SLACK_FMIN=$FM ADC_IRV_SLACK_R=$R timeout 100 python tools/gpu_k8_stats.py $SZ
    done
  done
  ADC_IRV_SLACK=0 timeout 100 python tools/gpu_k8_stats.py $SZ
done
