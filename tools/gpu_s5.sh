#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/s5_pytest.log; cat $O/s5_pytest.log
timeout 900 python bench.py --workload structured --steps 10 --no-cpu-baseline > $O/s5_bench_structured.json 2> $O/s5_bench_structured.err; echo "structured rc=$?"
python - <<'PY'
import json
for n in ["s5_bench_structured"]:
    try:
        o=json.loads(open("gpurun_out/%s.json"%n).read().strip().splitlines()[-1])
        print(n, o["value"], o["stage_ms"], "farm", o["farm_check"]["ok"], {k:(o[k]["value"]) for k in ("structured","noise","host_inclusive","throughput_mode") if k in o})
    except Exception as e: print(n, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
for CFG in "structured 1920 1080"; do
  set -- $CFG; WL=$1; W=$2; H=$3; TAG=${WL}_${W}x${H}
  rm -rf "$REPO/$O/prof5_$TAG"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof5_$TAG" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --workload $WL --width $W --height $H > "$REPO/$O/rocprof5_$TAG.log" 2>&1; echo "rocprof $TAG rc=$?"
  (cd "$REPO"; python tools/prof_summary.py $(ls $O/prof5_$TAG/*.db $O/prof5_$TAG/*/*.db 2>/dev/null | tail -1) > $O/s5_kernel_stats_$TAG.md 2>&1; head -8 $O/s5_kernel_stats_$TAG.md)
done
