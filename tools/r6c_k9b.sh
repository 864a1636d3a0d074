#!/bin/bash
# round 6, second session: K9 on the code map -- variants (finished rays masked out of the gathers, 4 / 6 / 8 steps per trip), same box,
# interleaved; kernel table of the default library.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out; L=$GRAFT_REPO_ROOT/adcensus_amd/lib
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
for rep in 1 2; do
  ARGS="--steps 20 $B --workload noise"
  run k9b_noise_base_$rep ADC_HIP_LIB=$L/r6base/libadcensus_hip.so
  run k9b_noise_new_$rep X=1
  for v in k9mask k9ns4 k9ns4mask k9ns6; do run k9b_noise_${v}_$rep ADC_HIP_LIB=$L/$v/libadcensus_hip.so; done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/$O/prof_k9b
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_k9b -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --workload noise > $R/$O/rocprof_k9b.log 2>&1; echo "rocprof rc=$?"
cd $R; python tools/prof_summary.py $(ls $O/prof_k9b/*.db $O/prof_k9b/*/*.db 2>/dev/null | tail -1) > $O/k9b_kernel_stats_noise.md 2>&1; head -40 $O/k9b_kernel_stats_noise.md | cut -c1-150
