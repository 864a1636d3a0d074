#!/bin/bash
# kernel tables of the current tree, both workloads (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $R/$O
for WL in noise structured; do
  rm -rf $R/$O/prof_c_$WL
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c_$WL -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --workload $WL > $R/$O/rocprof_c_$WL.log 2>&1; echo "rocprof $WL rc=$?"
  (cd $R; python tools/prof_summary.py $(ls $O/prof_c_$WL/*.db $O/prof_c_$WL/*/*.db 2>/dev/null | tail -1) > $O/c_kernel_stats_$WL.md 2>&1; head -45 $O/c_kernel_stats_$WL.md | cut -c1-130)
done
