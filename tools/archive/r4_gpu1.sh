#!/bin/bash
# round 4, GPU call 1: yardstick ubench, GPU test tier (stop at the first failure), then same-box A/B of the new K8 list layout /
# one-trip votes and the K5 verified segments against the round-3 library (adcensus_amd/lib/r3, built from bedf0e8)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 120 tools/ubench/copy_ceiling > $O/r4_ubench_copy_ceiling.txt 2>&1; echo "ubench rc=$?"; tail -14 $O/r4_ubench_copy_ceiling.txt
timeout 540 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -22 > $O/r4_gpu_pytest_1.log; cat $O/r4_gpu_pytest_1.log
grep -q " passed" $O/r4_gpu_pytest_1.log && ! grep -q "failed\|error" $O/r4_gpu_pytest_1.log || { echo "TESTS NOT GREEN -- stopping"; exit 1; }
B="--no-cpu-baseline --no-extra-legs"
run() { # tag, env..., -- bench args
  TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4a_$TAG.json 2> $O/r4a_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4a_$TAG.json"))
    print("%-28s rc=$rc  %.1f pairs/s  %.3f ms  stages %s  fallbacks %s" % ("$TAG", d["value"], d["ms_per_step"], d.get("stage_ms"), d.get("async_fallbacks")))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
for rep in 1 2; do
  run struct_new_$rep X=1 -- --workload structured --steps 10
  run struct_r3_$rep ADC_HIP_LIB=$REPO/adcensus_amd/lib/r3/libadcensus_hip.so -- --workload structured --steps 10
  run struct_noseg_$rep ADC_SO_SEG=0 -- --workload structured --steps 10
done
run struct_grid256 ADC_IRV_GRID=256 -- --workload structured --steps 10
for rep in 1 2; do
  run noise_new_$rep X=1 -- --steps 20
  run noise_noseg_$rep ADC_SO_SEG=0 -- --steps 20
done
run noise_seg3 ADC_SO_SEG=3 -- --steps 20
run noise_seg4 ADC_SO_SEG=4 -- --steps 20
for WL in noise structured; do
  run kitti_${WL}_new X=1 -- --width 1242 --height 375 --workload $WL --steps 30
  run kitti_${WL}_noseg ADC_SO_SEG=0 -- --width 1242 --height 375 --workload $WL --steps 30
  run kitti_${WL}_seg3 ADC_SO_SEG=3 -- --width 1242 --height 375 --workload $WL --steps 30
  run kitti_${WL}_r3 ADC_HIP_LIB=$REPO/adcensus_amd/lib/r3/libadcensus_hip.so -- --width 1242 --height 375 --workload $WL --steps 30
done
cd /tmp && export TMPDIR=/tmp
TAG=structured_1920x1080
rm -rf "$REPO/$O/prof_$TAG"
timeout 120 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_$TAG" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 $B --workload structured > "$REPO/$O/rocprof_$TAG.log" 2>&1; echo "rocprof $TAG rc=$?"
cd "$REPO"
DB=$(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | tail -1)
python tools/prof_summary.py $DB > $O/r4a_kernel_stats_$TAG.md 2>&1; head -30 $O/r4a_kernel_stats_$TAG.md | cut -c1-150
timeout 120 python tools/irv_trace_summary.py $DB > $O/r4a_irv_chain_structured.txt 2>&1; head -8 $O/r4a_irv_chain_structured.txt
