#!/bin/bash
# GPU gate: parity suite, the two bench lines, the 2-rank farm on one GPU (gloo), rocprofv3 kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/gate_pytest.log; cat $O/gate_pytest.log
timeout 900 python bench.py > $O/gate_bench_default.json 2> $O/gate_bench_default.err; echo "default rc=$?"
timeout 900 python bench.py --workload structured --steps 10 --no-cpu-baseline > $O/gate_bench_structured.json 2> $O/gate_bench_structured.err; echo "structured rc=$?"
python - <<'PY'
import json
for n in ["gate_bench_default","gate_bench_structured"]:
    try:
        o=json.loads(open("gpurun_out/%s.json"%n).read().strip().splitlines()[-1])
        print(n, o["value"], o["stage_ms"], "farm", o["farm_check"]["ok"], {k:(o[k]["value"]) for k in ("structured","noise","host_inclusive","throughput_mode") if k in o})
    except Exception as e: print(n, "ERR", e)
PY
ADC_BENCH_BACKEND=gloo ADC_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > $O/gate_bench_2ranks_gloo.json 2> $O/gate_bench_2ranks_gloo.err; echo "2ranks rc=$?"; cut -c1-700 $O/gate_bench_2ranks_gloo.json
cd /tmp && export TMPDIR=/tmp
for CFG in "structured 1920 1080" "noise 1920 1080"; do
  set -- $CFG; WL=$1; W=$2; H=$3; TAG=${WL}_${W}x${H}
  rm -rf "$REPO/$O/prof_gate_$TAG"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_gate_$TAG" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --workload $WL --width $W --height $H > "$REPO/$O/rocprof_gate_$TAG.log" 2>&1; echo "rocprof $TAG rc=$?"
  (cd "$REPO"; python tools/prof_summary.py $(ls $O/prof_gate_$TAG/*.db $O/prof_gate_$TAG/*/*.db 2>/dev/null | tail -1) > $O/gate_kernel_stats_$TAG.md 2>&1; head -14 $O/gate_kernel_stats_$TAG.md)
done
