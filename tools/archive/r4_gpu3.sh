#!/bin/bash
# round 4, GPU call 3: K4 step-cost micro-benchmark; the banded median with one store sink per lane (speculative and chained
# forms): correctness (median tests + one full-size pair), then same-box A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 120 tools/ubench/k4_step_cost > $O/r4_ubench_k4_step_cost.txt 2>&1; echo "ubench rc=$?"; cat $O/r4_ubench_k4_step_cost.txt
timeout 400 python -m pytest tests/test_gpu_api.py tests/test_gpu_fullsize.py -m gpu -x -q -k "median or (full_size_match and noise-12345) or kitti or contract" 2>&1 | tail -6 > $O/r4_gpu_pytest_3.log; cat $O/r4_gpu_pytest_3.log
grep -q " passed" $O/r4_gpu_pytest_3.log && ! grep -q "failed\|error" $O/r4_gpu_pytest_3.log || { echo "TESTS NOT GREEN -- stopping"; exit 1; }
B="--no-cpu-baseline --no-extra-legs"
run() { # tag, env..., -- bench args
  TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4c_$TAG.json 2> $O/r4c_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4c_$TAG.json"))
    print("%-26s rc=$rc  %.1f pairs/s  %.3f ms  stages %s  fallbacks %s" % ("$TAG", d["value"], d["ms_per_step"], d.get("stage_ms"), {k: v for k, v in d.get("async_fallbacks", {}).items() if k in ("median_handoff",)}))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
for rep in 1 2; do
  run noise_spec2_$rep X=1 -- --steps 20
  run noise_chain_$rep ADC_MEDIAN_SPEC=0 -- --steps 20
  run noise_spec1_$rep ADC_MEDIAN_SPEC=1 -- --steps 20
done
run noise_spec3 ADC_MEDIAN_SPEC=3 -- --steps 20
run struct_spec2 X=1 -- --workload structured --steps 10
run struct_chain ADC_MEDIAN_SPEC=0 -- --workload structured --steps 10
run kitti_noise_spec2 X=1 -- --width 1242 --height 375 --steps 30
run kitti_noise_chain ADC_MEDIAN_SPEC=0 -- --width 1242 --height 375 --steps 30
cd /tmp && export TMPDIR=/tmp
for V in spec2 chain; do
  TAG=noise_$V
  rm -rf "$REPO/$O/prof_$TAG"
  if [ $V = chain ]; then export ADC_MEDIAN_SPEC=0; else unset ADC_MEDIAN_SPEC; fi
  timeout 120 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_$TAG" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 $B > "$REPO/$O/rocprof_$TAG.log" 2>&1; echo "rocprof $TAG rc=$?"
  (cd "$REPO"; python tools/prof_summary.py $(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | tail -1) > $O/r4c_kernel_stats_$TAG.md 2>&1; grep -i "median\|interpolate" $O/r4c_kernel_stats_$TAG.md | cut -c1-120)
done
