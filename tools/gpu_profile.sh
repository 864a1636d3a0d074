#!/bin/bash
# round-2 GPU session 3: new bench.py lines (default + structured + KITTI size), farm digests, 2-rank farm on one GPU,
# rocprofv3 kernel stats and PMC traffic passes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 600 python bench.py --steps 160 --warmup 3 --no-cpu-baseline --no-extra-legs --write-digests $O/farm_digests.json > $O/s3_bench_digests.json 2> $O/s3_bench_digests.err; echo "digests rc=$?"
cp $O/farm_digests.json tests/golden/farm_digests.json
timeout 900 python bench.py > $O/r2_bench_default.json 2> $O/r2_bench_default.err; echo "default rc=$?"; cut -c1-600 $O/r2_bench_default.json
timeout 900 python bench.py --workload structured --steps 10 > $O/r2_bench_structured.json 2> $O/r2_bench_structured.err; echo "structured rc=$?"
ADC_BENCH_BACKEND=gloo ADC_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/s3_bench_2ranks_gloo.json 2> $O/s3_bench_2ranks_gloo.err; echo "2ranks rc=$?"; cut -c1-900 $O/s3_bench_2ranks_gloo.json
for WL in noise structured; do
  timeout 600 python bench.py --width 1242 --height 375 --workload $WL --steps 20 --no-cpu-baseline --no-extra-legs > $O/r2_bench_kitti_$WL.json 2> $O/r2_bench_kitti_$WL.err; echo "kitti $WL rc=$?"
done
cd /tmp && export TMPDIR=/tmp
for CFG in "noise 1920 1080" "structured 1920 1080" "noise 1242 375" "structured 1242 375"; do
  set -- $CFG; WL=$1; W=$2; H=$3; TAG=${WL}_${W}x${H}
  rm -rf "$REPO/$O/prof_$TAG"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_$TAG" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs --workload $WL --width $W --height $H > "$REPO/$O/rocprof_$TAG.log" 2>&1; echo "rocprof $TAG rc=$?"
  (cd "$REPO"; python tools/prof_summary.py $(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | tail -1) > $O/r2_kernel_stats_$TAG.md 2>&1)
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$REPO/$O/pmc_${TAG}_$C"
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$REPO/$O/pmc_${TAG}_$C" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --workload $WL --width $W --height $H > "$REPO/$O/pmc_${TAG}_$C.log" 2>&1; echo "pmc $TAG $C rc=$?"
  done
done
cd "$REPO"; ls $O | grep -c pmc_; head -12 $O/r2_kernel_stats_structured_1920x1080.md
