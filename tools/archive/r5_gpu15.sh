#!/bin/bash
O=gpurun_out/r5_15; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_stages.py tests/test_gpu_api.py -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
timeout 600 python bench.py --workload structured --steps 10 --no-cpu-baseline > $O/bench_structured.json 2> $O/bench_structured.err; echo "bench rc=$?"
python - <<'P'
import json
o=json.load(open('gpurun_out/r5_15/bench_structured.json'))
print(o['value'], o['ms_per_step'], o['stage_ms'], o['farm_check']['reference_checked'], o['farm_check']['reference_mismatches'], o['async_fallbacks'])
print('thr', o['throughput_mode']['value'], 'mixed', o['mixed_stream']['value'], o['mixed_stream']['reference_mismatches'], 'noise leg', o['noise']['value'], o['noise']['stage_ms'])
P
