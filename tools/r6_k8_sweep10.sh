#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
timeout 900 python -m pytest tests/test_gpu_random.py tests/test_gpu_stages.py tests/test_gpu_api.py -m gpu -q 2>&1 | tail -3
for S in "1080 --steps 10" "kitti --width 1242 --height 375 --steps 30"; do
  set -- $S; T=$1; shift
  ARGS="$* $B --workload structured"
  for rep in 1 2; do
  for F in 1 4 8 16 32; do run k8h_${T}_f${F}_$rep ADC_IRV_SLACK=$F; done
  done
done
ARGS="--steps 20 $B --workload noise"; run k8h_noise X=1
cd /tmp && export TMPDIR=/tmp; REPO="$GRAFT_REPO_ROOT"
rm -rf "$REPO/$O/prof_k8"
timeout 200 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_k8" -o bench -- python "$REPO/bench.py" --steps 4 --warmup 2 $B --workload structured > "$REPO/$O/rocprof_k8.log" 2>&1
(cd "$REPO"; DB=$(ls $O/prof_k8/*.db $O/prof_k8/*/*.db 2>/dev/null | tail -1); python tools/irv_trace_summary.py $DB | head -5 | cut -c1-600)
rm -rf "$REPO/$O/prof_k8"
