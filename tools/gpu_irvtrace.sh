#!/bin/bash
# IRV round trace of the structured 1080p pair (ADC_IRV_TRACE): dirty-list length per round
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
ADC_IRV_TRACE=1 timeout 600 python - > gpurun_out/irv_trace.log 2>&1 <<'PY'
import numpy as np, adcensus_amd as A
from adcensus_amd import workloads
l, r = workloads.structured_pair(1920, 1080, 128, seed=777)
st = A.ADCensusStereo(device=0)
assert st.Initialize(1920, 1080, A.ADCensusOption(max_disparity=128))
d = np.empty((1080, 1920), np.float32)
assert st.Match(l, r, d)
PY
echo rc=$?; wc -l gpurun_out/irv_trace.log
