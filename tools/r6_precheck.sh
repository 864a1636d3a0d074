#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_random.py tests/test_gpu_stages.py tests/test_gpu_api.py tests/test_gpu_faults.py tests/test_reference_caller.py tests/test_gpu_fused_tail.py tests/test_gpu_paper.py -m gpu -q 2>&1 | tail -4
B="--no-cpu-baseline --no-extra-legs"
for rep in 1 2; do
timeout 100 python bench.py --steps 20 $B > gpurun_out/pre_noise_$rep.json 2>/dev/null; python tools/bench_brief.py gpurun_out/pre_noise_$rep.json
ADC_HIP_LIB=adcensus_amd/lib/r5/libadcensus_hip.so timeout 100 python bench.py --steps 20 $B > gpurun_out/pre_noise_r5_$rep.json 2>/dev/null; python tools/bench_brief.py gpurun_out/pre_noise_r5_$rep.json
done
