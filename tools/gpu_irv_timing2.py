"""Diagnosis (needs the IRV_TIMING variant: tools/build_variant.sh irvt "-DIRV_TIMING=1" k_voting.hip): the first rounds of the
region-voting chain, kernel by kernel, in the 8 sampled workgroups (0, 64, .., 448): duration of the workgroup's wave 0 (us), cycles
spent up to the end of phase 1 / of the pool barrier, pool items of the workgroup's first batch, votes wave 0 really took.
  ADC_HIP_LIB=adcensus_amd/lib/irvt/libadcensus_hip.so python tools/gpu_irv_timing2.py [kernels]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import adcensus_amd as A
from adcensus_amd import workloads

NK = int(sys.argv[1]) if len(sys.argv) > 1 else 24
W, H, D = 1920, 1080, 128
left, right = workloads.structured_pair(W, H, D)
st = A.ADCensusStereo(device=0)
assert st.Initialize(W, H, A.ADCensusOption(max_disparity=D))
out = np.empty((H, W), np.float32)
for i in range(3):
    assert st.Match(left, right, out)
L = A.lib()
L.adc_debug_irv_timing.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
print("slack", os.environ.get("ADC_IRV_SLACK", "1"), "fmin", os.environ.get("ADC_IRV_SLACK_FMIN"), "r", os.environ.get("ADC_IRV_SLACK_R"),
      "rounds/evals", st.voting_stats(), "budget", st.debug_counter(3))
N = 256
B = []
for which in range(8):
    buf = np.zeros((N, 12), np.int64)
    L.adc_debug_irv_timing(buf.ctypes.data, which, N)
    B.append(buf)
B = np.stack(B)
for k in range(NK):
    t = B[:, k, :]
    dur = (t[:, 9] - t[:, 8]) * 0.01  # 100 MHz
    skew = (t[:, 8] - t[:, 8].min()) * 0.01
    ph1 = t[:, 2] - t[:, 0]
    pool = t[:, 3] - t[:, 2]
    print("k %3d  wave-0 us %s | cyc to end of phase 1 %s  pool barrier %s | pool items %s  votes(wave 0) %s | LAST item of wave 0, cycles: loads issued -> first block back %s, gather + histogram %s, votes of the levels %s" % (
        k, np.round(dur, 1).tolist(), ph1.tolist(), pool.tolist(), t[:, 10].tolist(), t[:, 11].tolist(),
        (t[:, 5] - t[:, 4]).tolist(), (t[:, 6] - t[:, 5]).tolist(), (t[:, 7] - t[:, 6]).tolist()))
