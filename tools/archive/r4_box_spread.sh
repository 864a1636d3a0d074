#!/bin/bash
# round 4: the headline line (no CPU baseline, no extra legs) on whatever box this call lands on -- the spread of the pool
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-a}
timeout 120 python bench.py --no-cpu-baseline --no-extra-legs --steps 20 > gpurun_out/r4_box_$TAG.json 2> gpurun_out/r4_box_$TAG.err; echo "rc=$?"
timeout 120 python bench.py --no-cpu-baseline --no-extra-legs --steps 10 --workload structured > gpurun_out/r4_box_${TAG}_structured.json 2>> gpurun_out/r4_box_$TAG.err
python - <<PY
import json
for f in ("gpurun_out/r4_box_$TAG.json", "gpurun_out/r4_box_${TAG}_structured.json"):
    d = json.load(open(f)); r = d["roofline"]
    print(f, "%.1f pairs/s %.3f ms  K4 %.4f ms frac %.3f  copy %.0f GB/s  stages %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["device_copy_GBps"], d["stage_ms"]))
PY
