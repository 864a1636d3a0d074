#!/bin/bash
# round-3 GPU session 4: the new bench.py (real-bytes roofline, stage fractions, pull queue, host-fed leg), chunk A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
show() { python - "$1" <<'PY'
import json, sys
n = sys.argv[1]
try:
    o = json.loads(open("gpurun_out/%s.json" % n).read().strip().splitlines()[-1])
    r = o["roofline"]
    print(n, "value", o["value"], "ms", o["ms_per_step"], o["stage_ms"], "k4", r["avg_launch_ms"], "frac", r["frac"], "alg", r["algorithmic_frac"], "farm", o["farm_check"]["ok"], o["config"]["queue"][:6], o.get("scaling"))
    for k in ("structured", "noise"):
        if k in o: print("   ", k, o[k]["value"], o[k]["stage_ms"], o[k]["roofline"]["avg_launch_ms"], o[k]["roofline"]["frac"])
    for k in ("host_inclusive", "host_farm", "throughput_mode", "scaling_reference"):
        if k in o: print("   ", k, o[k]["value"])
    print("    stage_roofline", {k: v["frac"] for k, v in o["stage_roofline"].items() if isinstance(v, dict)})
except Exception as e:
    print(n, "ERR", e); print(open("gpurun_out/%s.err" % n).read()[-1500:])
PY
}
timeout 900 python bench.py > $O/g4_default.json 2> $O/g4_default.err; echo "default rc=$?"; show g4_default
timeout 600 python bench.py --workload structured --steps 10 --no-cpu-baseline --no-extra-legs > $O/g4_struct.json 2> $O/g4_struct.err; echo "struct rc=$?"; show g4_struct
ADC_AGG_HCHUNK=384 timeout 600 python bench.py --workload structured --steps 10 --no-cpu-baseline --no-extra-legs > $O/g4_struct_h384.json 2> $O/g4_struct_h384.err; show g4_struct_h384
ADC_AGG_SMALL_L=0 timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-extra-legs > $O/g4_noise_fullring.json 2> $O/g4_noise_fullring.err; show g4_noise_fullring
timeout 600 python bench.py --batch 16 --no-cpu-baseline --no-extra-legs > $O/g4_batch16.json 2> $O/g4_batch16.err; echo "batch16 rc=$?"; show g4_batch16
ADC_BENCH_BACKEND=gloo ADC_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/g4_2ranks_weak.json 2> $O/g4_2ranks_weak.err; echo "2ranks weak rc=$?"; show g4_2ranks_weak
ADC_BENCH_BACKEND=gloo ADC_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --batch 16 --warmup 2 > $O/g4_2ranks_batch16.json 2> $O/g4_2ranks_batch16.err; echo "2ranks batch rc=$?"; show g4_2ranks_batch16
ADC_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/g4_nccl1.json 2> $O/g4_nccl1.err; echo "nccl 1 rank rc=$?"; show g4_nccl1
