#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for MP in 1 0; do echo "ADC_MEDIAN_PAIRS=$MP"; ADC_MEDIAN_PAIRS=$MP timeout 300 python tools/gpu_median_probe.py; done 2>&1 | tee gpurun_out/g10_median_probe.txt
timeout 1500 python -m pytest tests/test_gpu_stages.py "tests/test_gpu_api.py::test_median_handoff_timeout_falls_back" "tests/test_gpu_fullsize.py::test_full_size_match_equals_reference" "tests/test_gpu_fullsize.py::test_kitti_size_match_equals_reference" -m gpu -x -q 2>&1 | tail -4
