#!/bin/bash
# round 4, GPU call 2: GPU test tier on the tree with the K4 batched decode + the speculative median bands, then same-box A/B:
# median (ADC_MEDIAN_SPEC 2 / 1 / 0), K4 (this tree vs adcensus_amd/lib/k4old = the tree before the batched decode), kernel tables
# and SQ counters of the structured pair.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 560 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -22 > $O/r4_gpu_pytest_2.log; cat $O/r4_gpu_pytest_2.log
grep -q " passed" $O/r4_gpu_pytest_2.log && ! grep -q "failed\|error" $O/r4_gpu_pytest_2.log || { echo "TESTS NOT GREEN -- stopping"; exit 1; }
B="--no-cpu-baseline --no-extra-legs"
run() { # tag, env..., -- bench args
  TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4b_$TAG.json 2> $O/r4b_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4b_$TAG.json"))
    print("%-26s rc=$rc  %.1f pairs/s  %.3f ms  stages %s  K4 launch %.4f frac %.3f  fallbacks %s" % ("$TAG", d["value"], d["ms_per_step"], d.get("stage_ms"), d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("async_fallbacks")))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
K4OLD=ADC_HIP_LIB=$REPO/adcensus_amd/lib/k4old/libadcensus_hip.so
for rep in 1 2; do
  run struct_new_$rep X=1 -- --workload structured --steps 10
  run struct_k4old_$rep $K4OLD -- --workload structured --steps 10
  run struct_medchain_$rep ADC_MEDIAN_SPEC=0 -- --workload structured --steps 10
  run noise_new_$rep X=1 -- --steps 20
  run noise_medchain_$rep ADC_MEDIAN_SPEC=0 -- --steps 20
  run noise_k4old_$rep $K4OLD -- --steps 20
done
run noise_medspec1 ADC_MEDIAN_SPEC=1 -- --steps 20
run noise_medspec3 ADC_MEDIAN_SPEC=3 -- --steps 20
run noise_fullring ADC_AGG_SMALL_L=0 -- --steps 20
run noise_fullring_k4old ADC_AGG_SMALL_L=0 $K4OLD -- --steps 20
for WL in noise structured; do
  run kitti_${WL}_new X=1 -- --width 1242 --height 375 --workload $WL --steps 30
  run kitti_${WL}_medchain ADC_MEDIAN_SPEC=0 -- --width 1242 --height 375 --workload $WL --steps 30
  run kitti_${WL}_k4old $K4OLD -- --width 1242 --height 375 --workload $WL --steps 30
done
cd /tmp && export TMPDIR=/tmp
for WL in noise structured; do
  TAG=${WL}_1920x1080
  rm -rf "$REPO/$O/prof_$TAG"
  timeout 120 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_$TAG" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 $B --workload $WL > "$REPO/$O/rocprof_$TAG.log" 2>&1; echo "rocprof $TAG rc=$?"
  (cd "$REPO"; python tools/prof_summary.py $(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | tail -1) > $O/r4b_kernel_stats_$TAG.md 2>&1; head -16 $O/r4b_kernel_stats_$TAG.md | cut -c1-120)
done
WL=structured; i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rm -rf "$REPO/$O/pmcsq_${WL}_$i"
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$REPO/$O/pmcsq_${WL}_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 $B --workload $WL > "$REPO/$O/pmcsq_${WL}_$i.log" 2>&1; echo "sq $WL pass $i rc=$?"
done
(cd "$REPO"; python tools/pmc_sq_summary.py $O/pmcsq_${WL}_ > $O/r4b_sq_all_$WL.md 2>&1; grep -i "agg_rr2\|kernel" $O/r4b_sq_all_$WL.md | head -12 | cut -c1-220)
