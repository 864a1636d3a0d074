#!/bin/bash
# round-3 GPU session 2: parity suite with the pair-register ring (k_aggregate_rr2.h), A/B against the one-float ring
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/g2_pytest.log; cat $O/g2_pytest.log
for RR2 in 1 0; do
  ADC_AGG_RR2=$RR2 timeout 600 python bench.py --workload structured --steps 10 --no-cpu-baseline --no-extra-legs > $O/g2_bench_struct_rr2_$RR2.json 2> $O/g2_bench_struct_rr2_$RR2.err; echo "structured rr2=$RR2 rc=$?"
done
python - <<'PY'
import json
for n in ["g2_bench_struct_rr2_1","g2_bench_struct_rr2_0"]:
    try:
        o=json.loads(open("gpurun_out/%s.json"%n).read().strip().splitlines()[-1])
        print(n, o["value"], o["stage_ms"], o["roofline"]["avg_launch_ms"], o["roofline"]["hbm_frac"], "farm", o["farm_check"]["ok"])
    except Exception as e: print(n, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
for WL in structured; do
  rm -rf "$REPO/$O/prof_g2_$WL"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_g2_$WL" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --workload $WL > "$REPO/$O/rocprof_g2_$WL.log" 2>&1; echo "rocprof $WL rc=$?"
  (cd "$REPO"; python tools/prof_summary.py $(ls $O/prof_g2_$WL/*.db $O/prof_g2_$WL/*/*.db 2>/dev/null | tail -1) > $O/g2_kernel_stats_$WL.md 2>&1; head -16 $O/g2_kernel_stats_$WL.md)
done
