#!/bin/bash
# per-launch kernel trace of the IRV kernels of ONE structured 1080p Match (durations in launch order)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
R="$GRAFT_REPO_ROOT"
cat > /tmp/one.py <<'PY'
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import adcensus_amd as A
from adcensus_amd import workloads
l, r = workloads.structured_pair(1920, 1080, 128, seed=777)
st = A.ADCensusStereo(device=0)
assert st.Initialize(1920, 1080, A.ADCensusOption(max_disparity=128))
d = np.empty((1080, 1920), np.float32)
for _ in range(2): assert st.Match(l, r, d)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/irvprof
ADC_IRV_TRACE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/irvprof -o t -- python /tmp/one.py "$R" > "$R/gpurun_out/irvprof.log" 2>&1
echo rc=$?
f=$(find /tmp/irvprof -name '*kernel_trace.csv' | head -1)
python - "$f" > "$R/gpurun_out/irv_launches.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    prev_end = e
    print("%-40s dur %8.1f us  gap %7.1f us  grid %s" % (n[:40], (e - s) / 1e3, gap, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
PY
wc -l "$R/gpurun_out/irv_launches.txt"
