#!/bin/bash
# round-3 artefacts that depend on the K5 kernels, regenerated from the tree with the pinned prefetch slots: bench lines
# (default, structured, KITTI sizes) and the rocprofv3 kernel tables.  (The PMC traffic passes are keyed by the hash of the K4
# sources, which did not change; farm digests: outputs are bit-identical.)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 400 python bench.py > $O/r3_bench_default.json 2> $O/r3_bench_default.err; echo "default rc=$?"; cut -c1-300 $O/r3_bench_default.json
timeout 200 python bench.py --workload structured --steps 10 > $O/r3_bench_structured.json 2> $O/r3_bench_structured.err; echo "structured rc=$?"
for WL in noise structured; do
  timeout 120 python bench.py --width 1242 --height 375 --workload $WL --steps 20 --no-cpu-baseline --no-extra-legs > $O/r3_bench_kitti_$WL.json 2> $O/r3_bench_kitti_$WL.err; echo "kitti $WL rc=$?"
done
cd /tmp && export TMPDIR=/tmp
for CFG in "noise 1920 1080" "structured 1920 1080" "noise 1242 375" "structured 1242 375"; do
  set -- $CFG; WL=$1; W=$2; H=$3; TAG=${WL}_${W}x${H}
  rm -rf "$REPO/$O/prof_$TAG"
  timeout 150 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_$TAG" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs --workload $WL --width $W --height $H > "$REPO/$O/rocprof_$TAG.log" 2>&1; echo "rocprof $TAG rc=$?"
  (cd "$REPO"; python tools/prof_summary.py $(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | tail -1) > $O/r3_kernel_stats_$TAG.md 2>&1; grep scanline $O/r3_kernel_stats_$TAG.md | cut -c1-110)
done
