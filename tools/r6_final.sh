#!/bin/bash
# round 6, final artefacts from ONE tree in one gpurun call: GPU test tier; PMC traffic of the aggregation launches FIRST (separate
# FETCH_SIZE / WRITE_SIZE passes, both workloads, 1080p and the KITTI size) so that the bench lines behind it carry `roofline.traffic`;
# bench lines (default with CPU baselines / cone / extra legs, structured incl. its CPU baseline, KITTI sizes, fixed batch of 64 through
# the pull queue, two ranks on one GPU over gloo, RCCL with one rank); the C++ farm example; the mixed-stream stress; rocprofv3 kernel
# tables; SQ counters; the voting-chain trace.  Summaries -> gpurun_out/r6_*, copied into profiles/ afterwards.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -16 > $O/r6_gpu_pytest.log; cat $O/r6_gpu_pytest.log
grep -q " passed" $O/r6_gpu_pytest.log && ! grep -q "failed\|error" $O/r6_gpu_pytest.log || { echo "TESTS NOT GREEN -- stopping"; exit 1; }
B="--no-cpu-baseline --no-extra-legs"
cd /tmp && export TMPDIR=/tmp
for CFG in "noise 1920 1080" "structured 1920 1080" "noise 1242 375" "structured 1242 375"; do
  set -- $CFG; WL=$1; W=$2; H=$3; TAG=${WL}_${W}x${H}
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$REPO/$O/pmc_${TAG}_$C"
    timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$REPO/$O/pmc_${TAG}_$C" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 $B --workload $WL --width $W --height $H > "$REPO/$O/pmc_${TAG}_$C.log" 2>&1; echo "pmc $TAG $C rc=$?"
  done
  SUF=""; [ "$W" != "1920" ] && SUF="_${W}x${H}"
  (cd "$REPO"; python tools/pmc_summary.py $TAG > $O/r6_k4_pmc_traffic_$WL$SUF.json 2> $O/pmc_summary_$TAG.err; cp $O/r6_k4_pmc_traffic_$WL$SUF.json profiles/; head -c 300 $O/r6_k4_pmc_traffic_$WL$SUF.json; echo)
  rm -rf "$REPO/$O/pmc_${TAG}_FETCH_SIZE" "$REPO/$O/pmc_${TAG}_WRITE_SIZE"
done
cd "$REPO"
timeout 500 python bench.py > $O/r6_bench_default.json 2> $O/r6_bench_default.err; echo "default rc=$?"; python tools/bench_brief.py $O/r6_bench_default.json
timeout 500 python bench.py --workload structured --steps 10 --cpu-baseline-structured > $O/r6_bench_structured.json 2> $O/r6_bench_structured.err; echo "structured rc=$?"; python tools/bench_brief.py $O/r6_bench_structured.json
for WL in noise structured; do
  timeout 100 python bench.py --width 1242 --height 375 --workload $WL --steps 30 $B > $O/r6_bench_kitti_$WL.json 2> $O/r6_bench_kitti_$WL.err; echo "kitti $WL rc=$?"; python tools/bench_brief.py $O/r6_bench_kitti_$WL.json
done
timeout 200 python bench.py --batch 64 $B > $O/r6_bench_batch64_1gpu.json 2> $O/r6_bench_batch64_1gpu.err; echo "batch64 rc=$?"
ADC_BENCH_BACKEND=gloo ADC_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/r6_bench_2ranks_gloo_one_gpu.json 2> $O/r6_bench_2ranks_gloo.err; echo "2ranks rc=$?"
ADC_BENCH_FORCE_DIST=1 timeout 200 python bench.py --steps 6 --warmup 2 $B > $O/r6_bench_rccl_1rank.json 2> $O/r6_bench_rccl_1rank.err; echo "rccl 1 rank rc=$?"
timeout 200 adcensus_amd/bin/adcensus_farm_multi 32 > $O/r6_farm_multi_cpp_1gpu.json 2> $O/r6_farm_multi_cpp.err; echo "farm_multi rc=$?"; cat $O/r6_farm_multi_cpp_1gpu.json
timeout 200 python tools/gpu_stress_mixed.py 4 > $O/r6_stress_mixed.txt 2>&1; echo "stress rc=$?"; tail -2 $O/r6_stress_mixed.txt
cd /tmp
for CFG in "noise 1920 1080" "structured 1920 1080" "noise 1242 375" "structured 1242 375"; do
  set -- $CFG; WL=$1; W=$2; H=$3; TAG=${WL}_${W}x${H}
  rm -rf "$REPO/$O/prof_$TAG"
  timeout 150 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_$TAG" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 $B --workload $WL --width $W --height $H > "$REPO/$O/rocprof_$TAG.log" 2>&1; echo "rocprof $TAG rc=$?"
  (cd "$REPO"; python tools/prof_summary.py $(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | tail -1) > $O/r6_kernel_stats_$TAG.md 2>&1)
done
for WL in noise structured; do
  i=0
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    rm -rf "$REPO/$O/pmcsq_${WL}_$i"
    timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$REPO/$O/pmcsq_${WL}_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 $B --workload $WL > "$REPO/$O/pmcsq_${WL}_$i.log" 2>&1; echo "sq $WL pass $i rc=$?"
  done
  (cd "$REPO"; python tools/pmc_sq_summary.py $O/pmcsq_${WL}_ > $O/r6_sq_all_$WL.md 2>&1)
  rm -rf "$REPO/$O"/pmcsq_${WL}_*
done
cd "$REPO"; timeout 200 python tools/irv_trace_summary.py $(ls $O/prof_structured_1920x1080/*.db $O/prof_structured_1920x1080/*/*.db 2>/dev/null | tail -1) > $O/r6_irv_chain_structured.txt 2>&1
rm -rf $O/prof_*_1920x1080 $O/prof_*_1242x375
head -16 $O/r6_kernel_stats_structured_1920x1080.md | cut -c1-130; head -14 $O/r6_kernel_stats_noise_1920x1080.md | cut -c1-130; head -5 $O/r6_irv_chain_structured.txt | cut -c1-400
python - <<'P'
import json
for n in ("default", "structured", "kitti_noise", "kitti_structured"):
    o = json.load(open("gpurun_out/r6_bench_%s.json" % n)); r = o["roofline"]
    print(n, o["value"], "frac", r["frac"], "traffic", r["traffic"], r.get("traffic_over_bytes"), o["farm_check"]["ok"])
P
