#!/bin/bash
# round 5, call 6: K8 visibility experiments -- rounds / evaluations / chain time of the structured pair per variant
O=gpurun_out/r5_6; mkdir -p $O
cat > /tmp/irv_probe.py <<'P'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, adcensus_amd as A, hashlib
from adcensus_amd import workloads
W,H,D=1920,1080,128
l,r=workloads.structured_pair(W,H,D,seed=777)
st=A.ADCensusStereo(device=0); assert st.Initialize(W,H,A.ADCensusOption(max_disparity=D))
st.set_profiling(True)
for _ in range(4): out=st.match(l,r)
ms=[]
for _ in range(6):
    out=st.match(l,r); ms.append(st.stage_ms()["refine"])
print(sys.argv[1], "refine ms %.3f" % float(np.mean(ms)), "voting (rounds, evals)", st.voting_stats(), "budget", st.debug_counter(3), hashlib.sha256(out.tobytes()).hexdigest()[:12], flush=True)
P
for V in "XCD=1:UNC=0" "XCD=0:UNC=0" "XCD=0:UNC=1" "XCD=0:UNC=2" "XCD=1:UNC=1"; do
  X=${V%%:*}; U=${V##*:}
  ADC_IRV_XCD=${X##*=} ADC_IRV_UNCACHED=${U##*=} timeout 300 python /tmp/irv_probe.py "$V" 2>&1 | tail -1 | tee -a $O/irv_visibility.txt
done
