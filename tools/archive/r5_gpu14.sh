#!/bin/bash
# round 5: K8 workgroup shape (waves per workgroup x workgroups), 1080p and KITTI size, structured pairs
O=gpurun_out/r5_14; mkdir -p $O
cat > /tmp/irv_probe.py <<'P'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, adcensus_amd as A, hashlib
from adcensus_amd import workloads
W,H,D=int(sys.argv[2]),int(sys.argv[3]),128
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
print(sys.argv[1], "%dx%d" % (W, H), " | ".join(res), flush=True)
P
for rep in 1 2; do
for E in "ADC_IRV_WPB=16 ADC_IRV_GRID=512" "ADC_IRV_WPB=8 ADC_IRV_GRID=1024" "ADC_IRV_WPB=8 ADC_IRV_GRID=768" "ADC_IRV_WPB=8 ADC_IRV_GRID=1536" "ADC_IRV_WPB=8 ADC_IRV_GRID=2048" "ADC_IRV_WPB=4 ADC_IRV_GRID=1024"; do
  env $E timeout 300 python /tmp/irv_probe.py "$E" 1920 1080 2>&1 | tail -1 | tee -a $O/irv_shapes.txt
done
for E in "ADC_IRV_WPB=16 ADC_IRV_GRID=256" "ADC_IRV_WPB=8 ADC_IRV_GRID=512" "ADC_IRV_WPB=8 ADC_IRV_GRID=256" "ADC_IRV_WPB=8 ADC_IRV_GRID=1024" "ADC_IRV_WPB=4 ADC_IRV_GRID=512"; do
  env $E timeout 300 python /tmp/irv_probe.py "$E" 1242 375 2>&1 | tail -1 | tee -a $O/irv_shapes.txt
done
done
