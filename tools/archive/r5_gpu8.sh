#!/bin/bash
# round 5, call 8: K8 (bands of 8 rows, lazy histogram clear): all GPU tests + structured bench + probe
O=gpurun_out/r5_8; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/pytest_gpu.log
cat > /tmp/irv_probe.py <<'P'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, adcensus_amd as A, hashlib
from adcensus_amd import workloads
W,H,D=1920,1080,128
res=[]
for seed in (777, 779):
    l,r=workloads.structured_pair(W,H,D,seed=seed)
    st=A.ADCensusStereo(device=0); assert st.Initialize(W,H,A.ADCensusOption(max_disparity=D))
    st.set_profiling(True)
    for _ in range(4): out=st.match(l,r)
    ms=[]
    for _ in range(6):
        out=st.match(l,r); ms.append(st.stage_ms()["refine"])
    res.append("seed %d refine ms %.3f voting %s %s" % (seed, float(np.mean(ms)), st.voting_stats(), hashlib.sha256(out.tobytes()).hexdigest()[:8]))
    st.Release()
print(sys.argv[1], " | ".join(res), flush=True)
P
timeout 300 python /tmp/irv_probe.py "lazy clear" 2>&1 | tail -1 | tee $O/irv_probe.txt
timeout 600 python bench.py --workload structured --steps 10 --no-cpu-baseline > $O/bench_structured.json 2> $O/bench_structured.err; echo "bench rc=$?"
python - <<'P'
import json
o=json.load(open('gpurun_out/r5_8/bench_structured.json'))
print(o['value'], o['ms_per_step'], o['stage_ms'], o['farm_check']['reference_checked'], o['farm_check']['reference_mismatches'], o['async_fallbacks'])
print('thr', o['throughput_mode']['value'], 'mixed', o['mixed_stream']['value'], o['mixed_stream']['reference_mismatches'], 'noise leg', o['noise']['value'])
P
