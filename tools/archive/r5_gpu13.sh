#!/bin/bash
# round 5: K8 list-order parameters around the adopted point (bands of 8 rows, one column per tile), structured pairs
O=gpurun_out/r5_13; mkdir -p $O
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
for rep in 1 2; do
for V in base irv_b8t2 irv_b8t4 irv_b6 irv_b12; do
  L=adcensus_amd/lib/$V/libadcensus_hip.so; [ $V = base ] && L=adcensus_amd/lib/libadcensus_hip.so
  ADC_HIP_LIB=$L timeout 300 python /tmp/irv_probe.py "$V" 2>&1 | tail -1 | tee -a $O/irv_params2.txt
done
for E in "ADC_IRV_WPB=8 ADC_IRV_GRID=1024" "ADC_IRV_WPB=4 ADC_IRV_GRID=2048"; do
  env $E timeout 300 python /tmp/irv_probe.py "base $E" 2>&1 | tail -1 | tee -a $O/irv_params2.txt
done
done
