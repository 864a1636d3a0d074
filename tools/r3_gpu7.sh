#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "registered or async or farm" 2>&1 | tail -5
for HD in 0 1; do
ADC_HOST_DIRECT=$HD timeout 600 python bench.py --steps 10 --no-cpu-baseline > $O/g7_default_hd$HD.json 2> $O/g7_default_hd$HD.err; echo "rc=$?"
python - $HD <<'PY'
import json, sys
o = json.loads(open("gpurun_out/g7_default_hd%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("HOST_DIRECT", sys.argv[1], "value", o["value"], {k: o[k]["value"] for k in ("host_inclusive", "host_inclusive_registered", "host_farm", "throughput_mode")})
PY
done
