#!/bin/bash
# round 5, call 7: K8 list-order parameters (band height, tile width, cache policy of the state gathers), structured pair
O=gpurun_out/r5_7; mkdir -p $O
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
for V in irv_b8 irv_b4 irv_b2 irv_b1; do
  L=adcensus_amd/lib/$V/libadcensus_hip.so
  ADC_HIP_LIB=$L timeout 300 python /tmp/irv_probe.py "$V" 2>&1 | tail -1 | tee -a $O/irv_params.txt
done
for E in "ADC_IRV_WPB=8" "ADC_IRV_GRID=256" "ADC_IRV_GRID=1024" "ADC_IRV_XCD=0"; do
  env $E ADC_HIP_LIB=adcensus_amd/lib/irv_b8/libadcensus_hip.so timeout 300 python /tmp/irv_probe.py "irv_b8 $E" 2>&1 | tail -1 | tee -a $O/irv_params.txt
done
