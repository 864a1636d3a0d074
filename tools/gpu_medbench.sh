#!/bin/bash
# timing experiments on the banded median alone (ADC_MEDB_DBG bits: 1 no publish wait, 2 no take wait, 4 no polling)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for dbg in 0; do
ADC_MEDB_DBG=$dbg timeout 300 python - <<'PY'
import os, time, numpy as np, adcensus_amd as A
from adcensus_amd import workloads
l, r = workloads.noise_pair(1920, 1080, 12345)
st = A.ADCensusStereo(device=0)
assert st.Initialize(1920, 1080, A.ADCensusOption(max_disparity=128))
d = np.empty((1080, 1920), np.float32)
assert st.Match(l, r, d)
lib = A.lib()
for _ in range(3): st.debug_run(A.RUN_MEDIAN)
lib.adc_device_synchronize()
t = time.perf_counter()
N = 20
for _ in range(N): st.debug_run(A.RUN_MEDIAN)
lib.adc_device_synchronize()
print("dbg", os.environ["ADC_MEDB_DBG"], "median ms/run %.3f" % ((time.perf_counter() - t) / N * 1e3))
PY
done 2>&1 | grep "^dbg\|fault"
