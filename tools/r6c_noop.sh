#!/bin/bash
# no-op cost of the voting chain's kernel with 8 waves / 1 wave per workgroup (noise pair: BEGIN + DONE + surplus)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $R/$O
for WPB in 8 1 2; do
  rm -rf $R/$O/prof_noop_$WPB
  ADC_IRV_WPB=$WPB ADC_IRV_BUDGET=24 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_noop_$WPB -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --workload noise > $R/$O/rocprof_noop_$WPB.log 2>&1
  (cd $R; echo "WPB $WPB"; python tools/irv_trace_summary.py $(ls $O/prof_noop_$WPB/*.db $O/prof_noop_$WPB/*/*.db 2>/dev/null | tail -1) | head -4 | cut -c1-300)
  rm -rf $R/$O/prof_noop_$WPB
done
