#!/bin/bash
# round-3 final artefacts (one gpurun call, all from the same tree): bench lines (default, structured, KITTI sizes, fixed batch of
# 64 through the pull queue, 2-rank farms on one GPU over gloo, RCCL with one rank), farm digests, rocprofv3 kernel stats and
# PMC traffic passes for both workloads at 1080p, SQ counters.  Summaries -> gpurun_out/r3_*, copied into profiles/ afterwards.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 600 python bench.py --steps 160 --warmup 3 --no-cpu-baseline --no-extra-legs --write-digests $O/farm_digests.json > $O/r3_bench_digests.json 2> $O/r3_bench_digests.err; echo "digests rc=$?"
cp $O/farm_digests.json tests/golden/farm_digests.json
timeout 900 python bench.py > $O/r3_bench_default.json 2> $O/r3_bench_default.err; echo "default rc=$?"; cut -c1-400 $O/r3_bench_default.json
timeout 900 python bench.py --workload structured --steps 10 > $O/r3_bench_structured.json 2> $O/r3_bench_structured.err; echo "structured rc=$?"
timeout 600 python bench.py --batch 64 --no-cpu-baseline --no-extra-legs > $O/r3_bench_batch64_1gpu.json 2> $O/r3_bench_batch64_1gpu.err; echo "batch64 rc=$?"
ADC_BENCH_BACKEND=gloo ADC_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/r3_bench_2ranks_gloo_one_gpu.json 2> $O/r3_bench_2ranks_gloo.err; echo "2ranks rc=$?"
ADC_BENCH_BACKEND=gloo ADC_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --batch 32 --warmup 2 > $O/r3_bench_2ranks_gloo_batch32_one_gpu.json 2> $O/r3_bench_2ranks_gloo_b.err; echo "2ranks batch rc=$?"
ADC_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/r3_bench_rccl_1rank.json 2> $O/r3_bench_rccl_1rank.err; echo "rccl 1 rank rc=$?"
for WL in noise structured; do
  timeout 600 python bench.py --width 1242 --height 375 --workload $WL --steps 20 --no-cpu-baseline --no-extra-legs > $O/r3_bench_kitti_$WL.json 2> $O/r3_bench_kitti_$WL.err; echo "kitti $WL rc=$?"
done
cd /tmp && export TMPDIR=/tmp
for CFG in "noise 1920 1080" "structured 1920 1080" "noise 1242 375" "structured 1242 375"; do
  set -- $CFG; WL=$1; W=$2; H=$3; TAG=${WL}_${W}x${H}
  rm -rf "$REPO/$O/prof_$TAG"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_$TAG" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs --workload $WL --width $W --height $H > "$REPO/$O/rocprof_$TAG.log" 2>&1; echo "rocprof $TAG rc=$?"
  (cd "$REPO"; python tools/prof_summary.py $(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | tail -1) > $O/r3_kernel_stats_$TAG.md 2>&1)
done
for CFG in "noise 1920 1080" "structured 1920 1080"; do
  set -- $CFG; WL=$1; W=$2; H=$3; TAG=${WL}_${W}x${H}
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$REPO/$O/pmc_${TAG}_$C"
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$REPO/$O/pmc_${TAG}_$C" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --workload $WL --width $W --height $H > "$REPO/$O/pmc_${TAG}_$C.log" 2>&1; echo "pmc $TAG $C rc=$?"
  done
  (cd "$REPO"; python tools/pmc_summary.py $TAG > $O/r3_k4_pmc_traffic_$WL.json 2> $O/pmc_summary_$WL.err; head -c 600 $O/r3_k4_pmc_traffic_$WL.json)
done
for WL in noise structured; do
  i=0
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    rm -rf "$REPO/$O/pmcsq_${WL}_$i"
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$REPO/$O/pmcsq_${WL}_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --workload $WL > "$REPO/$O/pmcsq_${WL}_$i.log" 2>&1; echo "sq $WL pass $i rc=$?"
  done
  (cd "$REPO"; python tools/pmc_sq_summary.py $O/pmcsq_${WL}_ > $O/r3_sq_all_$WL.md 2>&1)
done
cd "$REPO"; timeout 300 python tools/irv_trace_summary.py $(ls $O/prof_structured_1920x1080/*.db $O/prof_structured_1920x1080/*/*.db 2>/dev/null | tail -1) > $O/r3_irv_chain_structured.txt 2>&1
head -14 $O/r3_kernel_stats_structured_1920x1080.md; head -12 $O/r3_kernel_stats_noise_1920x1080.md
