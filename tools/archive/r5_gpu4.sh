#!/bin/bash
# round 5, call 4: where does the joint voting chain spend its time?  kernel trace of the structured pair
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out/r5_4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extra-legs"
rm -rf "$REPO/$O/prof_structured"
timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_structured" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 $B --workload structured > "$REPO/$O/rocprof_structured.log" 2>&1; echo "rocprof rc=$?"
cd "$REPO"
DB=$(ls $O/prof_structured/*.db $O/prof_structured/*/*.db 2>/dev/null | tail -1)
python tools/prof_summary.py $DB > $O/kernel_stats_structured.md 2>&1; head -14 $O/kernel_stats_structured.md | cut -c1-140
python tools/irv_trace_summary.py $DB > $O/irv_chain_structured.txt 2>&1; cat $O/irv_chain_structured.txt | cut -c1-1500
python - <<'P'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, adcensus_amd as A
from adcensus_amd import workloads
W,H,D=1920,1080,128
l,r=workloads.structured_pair(W,H,D,seed=777)
st=A.ADCensusStereo(device=0); assert st.Initialize(W,H,A.ADCensusOption(max_disparity=D))
st.match(l,r); st.match(l,r)
print("voting stats (rounds, evals):", st.voting_stats(), "budget", st.debug_counter(3))
P
rm -rf $O/prof_structured
