#!/bin/bash
# round 5, call 1: two-plan aggregation (mixed streams), reference digests of the bench batches, first bench line
O=gpurun_out/r5_1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_api.py -x -q -k "assumptions or mixed_stream" > $O/pytest_api.log 2>&1; echo "api rc=$?"
tail -3 $O/pytest_api.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "digests" > $O/pytest_digests.log 2>&1; echo "digests rc=$?"
tail -3 $O/pytest_digests.log
timeout 600 python tools/gpu_stress_mixed.py 4 > $O/stress_mixed.txt 2>&1; echo "stress rc=$?"
tail -3 $O/stress_mixed.txt
ADC_AGG_DUAL=0 timeout 600 python tools/gpu_stress_mixed.py 4 > $O/stress_mixed_dual0.txt 2>&1; echo "stress(dual off) rc=$?"
tail -3 $O/stress_mixed_dual0.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
o=json.load(open('gpurun_out/r5_1/bench.json'))
print(o['value'], o['ms_per_step'], o['roofline']['frac'], o['farm_check'])
print('structured', o['structured']['value'], o['structured'].get('reference_check'), o['structured']['async_fallbacks'])
print('mixed', o.get('mixed_stream'))
print('thr', o['throughput_mode']['value'], 'stage', o['stage_ms'], o['structured']['stage_ms'])
P
