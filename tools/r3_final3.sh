#!/bin/bash
# final check of the tree with the two K5 kernel families: GPU test tier (stops everything at the first failure), then the
# artefacts that depend on K5 (bench lines + rocprofv3 kernel tables), then a same-box A/B of the 1080p scanline stage against
# the library with the round-2 scanline kernels (adcensus_amd/lib/so_old, built by hand from the previous revision)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 330 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14 > $O/r3_gpu_pytest.log; cat $O/r3_gpu_pytest.log
grep -q " passed" $O/r3_gpu_pytest.log && ! grep -q "failed\|error" $O/r3_gpu_pytest.log || { echo "TESTS NOT GREEN -- stopping"; exit 1; }
timeout 200 python bench.py > $O/r3_bench_default.json 2> $O/r3_bench_default.err; echo "default rc=$?"; cut -c1-260 $O/r3_bench_default.json
timeout 100 python bench.py --workload structured --steps 10 > $O/r3_bench_structured.json 2> $O/r3_bench_structured.err; echo "structured rc=$?"
for WL in noise structured; do
  timeout 60 python bench.py --width 1242 --height 375 --workload $WL --steps 20 --no-cpu-baseline --no-extra-legs > $O/r3_bench_kitti_$WL.json 2> $O/r3_bench_kitti_$WL.err; echo "kitti $WL rc=$?"
done
cd /tmp && export TMPDIR=/tmp
for CFG in "noise 1920 1080" "structured 1920 1080" "noise 1242 375" "structured 1242 375"; do
  set -- $CFG; WL=$1; W=$2; H=$3; TAG=${WL}_${W}x${H}
  rm -rf "$REPO/$O/prof_$TAG"
  timeout 100 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_$TAG" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs --workload $WL --width $W --height $H > "$REPO/$O/rocprof_$TAG.log" 2>&1; echo "rocprof $TAG rc=$?"
  (cd "$REPO"; python tools/prof_summary.py $(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | tail -1) > $O/r3_kernel_stats_$TAG.md 2>&1; grep scanline $O/r3_kernel_stats_$TAG.md | cut -c1-110)
done
cd "$REPO"
for rep in 1 2; do
 for V in old new; do
    if [ $V = old ]; then export ADC_HIP_LIB=$REPO/adcensus_amd/lib/so_old/libadcensus_hip.so; else unset ADC_HIP_LIB; fi
    timeout 60 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/k5d_${V}_$rep.json 2> $O/k5d_err.txt || break 2
    python -c "import json; d=json.load(open('$O/k5d_${V}_$rep.json')); print('$V', round(d['value'],1), d['stage_ms']['scanline'])"
 done
done 2>&1 | tee $O/r3_k5_ab_1080p.txt
