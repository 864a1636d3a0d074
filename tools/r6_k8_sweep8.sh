#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
L=adcensus_amd/lib
ARGS="--steps 20 $B --workload noise"
for rep in 1 2; do run k8f_n1080_r5lib_$rep ADC_HIP_LIB=$L/r5/libadcensus_hip.so; run k8f_n1080_slack1_$rep X=1; run k8f_n1080_slack0_$rep ADC_IRV_SLACK=0; done
ARGS="--steps 10 $B --workload structured"
for rep in 1 2; do run k8f_1080_r5lib_$rep ADC_HIP_LIB=$L/r5/libadcensus_hip.so; run k8f_1080_slack1_$rep X=1; run k8f_1080_slack0_$rep ADC_IRV_SLACK=0; done
cd /tmp && export TMPDIR=/tmp; REPO="$GRAFT_REPO_ROOT"
for V in new r5; do
  rm -rf "$REPO/$O/prof_n_$V"; E=""; [ $V = r5 ] && E="ADC_HIP_LIB=$REPO/$L/r5/libadcensus_hip.so"
  env $E timeout 200 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_n_$V" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 2 $B --workload noise > "$REPO/$O/rocprof_n_$V.log" 2>&1
  (cd "$REPO"; DB=$(ls $O/prof_n_$V/*.db $O/prof_n_$V/*/*.db 2>/dev/null | tail -1); python tools/prof_summary.py $DB > $O/r6f_kernel_stats_noise_$V.md 2>&1; head -30 $O/r6f_kernel_stats_noise_$V.md | cut -c1-110)
  rm -rf "$REPO/$O/prof_n_$V"
done
