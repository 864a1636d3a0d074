#!/bin/bash
# round 5, call 5: K8 with the tile-ordered (Gauss-Seidel) work list: stage tests, trace of the chain, bench line
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out/r5_5; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_stages.py -x -q > $O/pytest_stages.log 2>&1; echo "stages rc=$?"; tail -3 $O/pytest_stages.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "digests or isolation" > $O/pytest_fullsize.log 2>&1; echo "fullsize rc=$?"; tail -3 $O/pytest_fullsize.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extra-legs"
rm -rf "$REPO/$O/prof_structured"
timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_structured" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 $B --workload structured > "$REPO/$O/rocprof_structured.log" 2>&1; echo "rocprof rc=$?"
cd "$REPO"
DB=$(ls $O/prof_structured/*.db $O/prof_structured/*/*.db 2>/dev/null | tail -1)
python tools/prof_summary.py $DB > $O/kernel_stats_structured.md 2>&1; head -5 $O/kernel_stats_structured.md | cut -c1-140
python tools/irv_trace_summary.py $DB > $O/irv_chain_structured.txt 2>&1; cat $O/irv_chain_structured.txt | cut -c1-1500
rm -rf $O/prof_structured
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
timeout 600 python bench.py --workload structured --steps 10 --no-cpu-baseline --no-mixed-leg > $O/bench_structured.json 2> $O/bench_structured.err; echo "bench rc=$?"
python - <<'P'
import json
o=json.load(open('gpurun_out/r5_5/bench_structured.json'))
print(o['value'], o['ms_per_step'], o['stage_ms'], o['farm_check']['reference_checked'], o['farm_check']['reference_mismatches'], o['async_fallbacks'])
print('thr', o['throughput_mode']['value'])
P
