#!/bin/bash
# round 5, call 3: K8 as one fixed-point iteration over all ten passes -- stage tests, full-size tests, bench lines
O=gpurun_out/r5_3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_stages.py -x -q > $O/pytest_stages.log 2>&1; echo "stages rc=$?"; tail -3 $O/pytest_stages.log
timeout 1200 python -m pytest tests/test_gpu_api.py -x -q > $O/pytest_api.log 2>&1; echo "api rc=$?"; tail -3 $O/pytest_api.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q > $O/pytest_fullsize.log 2>&1; echo "fullsize rc=$?"; tail -3 $O/pytest_fullsize.log
timeout 600 python bench.py --workload structured --steps 10 --no-cpu-baseline > $O/bench_structured.json 2> $O/bench_structured.err; echo "bench rc=$?"
python - <<'P'
import json
o=json.load(open('gpurun_out/r5_3/bench_structured.json'))
print(o['value'], o['ms_per_step'], o['stage_ms'], o['farm_check']['reference_checked'], o['farm_check']['reference_mismatches'], o['async_fallbacks'])
print('thr', o['throughput_mode']['value'], 'mixed', o['mixed_stream']['value'], o['mixed_stream']['reference_mismatches'])
P
