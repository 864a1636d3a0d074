#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
for SL in 1 0 1 0; do ADC_IRV_SLACK=$SL timeout 120 python tools/gpu_k8_stats.py; done
for SL in 1 0; do ADC_IRV_SLACK=$SL timeout 120 python tools/gpu_k8_stats.py 1242 375; done
B="--no-cpu-baseline --no-extra-legs"
cd /tmp && export TMPDIR=/tmp
for SL in 1 0; do
  rm -rf "$REPO/$O/prof_k8_$SL"
  ADC_IRV_SLACK=$SL timeout 200 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_k8_$SL" -o bench -- python "$REPO/bench.py" --steps 4 --warmup 2 $B --workload structured > "$REPO/$O/rocprof_k8_$SL.log" 2>&1; echo "rocprof slack=$SL rc=$?"
  (cd "$REPO"; DB=$(ls $O/prof_k8_$SL/*.db $O/prof_k8_$SL/*/*.db 2>/dev/null | tail -1); python tools/irv_trace_summary.py $DB > $O/r6e_irv_chain_structured_slack$SL.txt 2>&1; head -8 $O/r6e_irv_chain_structured_slack$SL.txt | cut -c1-700)
  rm -rf "$REPO/$O/prof_k8_$SL"
done
