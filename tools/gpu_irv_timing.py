"""Diagnosis (needs the IRV_TIMING variant: tools/build_variant.sh irvt "-DIRV_TIMING=1" k_voting.hip):
where does a ROUND kernel of the region-voting chain spend its time?  Wave 0 of workgroups 0 and 1 stamp s_memtime at the
stage boundaries (drained loads) and s_memrealtime at kernel entry / exit.
  ADC_HIP_LIB=adcensus_amd/lib/irvt/libadcensus_hip.so python tools/gpu_irv_timing.py [W H]"""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.getcwd())
import adcensus_amd as A
from adcensus_amd import workloads
W, H, D = (int(sys.argv[1]), int(sys.argv[2]), 128) if len(sys.argv) > 2 else (1920, 1080, 128)
left, right = workloads.structured_pair(W, H, D)
st = A.ADCensusStereo(device=0)
opt = A.ADCensusOption(); opt.min_disparity = 0; opt.max_disparity = D
assert st.Initialize(W, H, opt)
out = np.empty((H, W), np.float32)
for i in range(3):
    assert st.Match(left, right, out)
L = A.lib()
N = 1200
L.adc_debug_irv_timing.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
print("rounds/evals", st.voting_stats(), "budget", st.debug_counter(3))
NS = 8
bufs = []
for which in range(8):
    buf = np.zeros((N, 10), np.int64)
    L.adc_debug_irv_timing(buf.ctypes.data, which, N)
    bufs.append(buf)
B = np.stack(bufs)  # [workgroup sample][kernel][slot]
ks = [k for k in range(30, N - 1) if (B[:, k, 8] > 0).all() and (B[:, k, 9] > B[:, k, 8]).all() and (B[:, k + 1, 8] > 0).all()]
print("ROUND kernels with complete stamps in all 8 sampled workgroups:", len(ks))
st0 = B[:, ks, 8].min(axis=0) * 0.01
first_exit = B[:, ks, 9].min(axis=0) * 0.01 - st0
last_start = B[:, ks, 8].max(axis=0) * 0.01 - st0
last_exit = B[:, ks, 9].max(axis=0) * 0.01 - st0
nxt = B[:, [k + 1 for k in ks], 8].min(axis=0) * 0.01 - st0
med = lambda a: float(np.median(a))
print("per kernel, relative to the earliest sampled entry (us, medians): latest entry %.2f, earliest exit %.2f, latest exit %.2f, next kernel's earliest entry %.2f (mean %.2f)"
      % (med(last_start), med(first_exit), med(last_exit), med(nxt), float(nxt.mean())))
for which in (0, 4):
    buf = B[which]
    cyc = buf[ks, :NS].astype(np.float64)
    d = np.diff(cyc, axis=1)
    ok = (d > 0).all(axis=1) & (d < 1e6).all(axis=1)
    if ok.any():
        print("workgroup %d, wave 0: complete stage stamps %d, median cycles [plan+entries, phase 1, pool, row arms, region, histogram, reduce]:" % (64 * which, int(ok.sum())),
              np.round(np.median(d[ok], axis=0), 0), " total", round(float(np.median(cyc[ok, NS - 1] - cyc[ok, 0]))))
