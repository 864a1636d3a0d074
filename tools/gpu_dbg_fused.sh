#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export PYTHONUNBUFFERED=1
timeout 300 python - <<'PY'
import numpy as np, adcensus_amd as A, sys
sys.path.insert(0, ".")
from tests import cases
left, right = cases.cone_pair()
opt = A.ADCensusOption(max_disparity=16)
h, w = left.shape[:2]
st = A.ADCensusStereo(device=0)
assert st.Initialize(w, h, opt)
st.debug_set_images(left, right)
st.debug_run(A.RUN_GRAY_CENSUS); st.debug_run(A.RUN_COST); st.debug_run(A.RUN_ARMS)
arms0 = np.zeros_like(st.debug_read(A.BUF_ARMS))
st.debug_write(A.BUF_ARMS, arms0)
st.debug_write(A.BUF_SUPCOUNT_H, np.ones_like(st.debug_read(A.BUF_SUPCOUNT_H)))
st.debug_write(A.BUF_SUPCOUNT_V, np.ones_like(st.debug_read(A.BUF_SUPCOUNT_V)))
st.debug_run(A.RUN_AGGREGATE, 1)
a = st.debug_read(A.BUF_VOLUME_A).copy()
st.debug_run(A.RUN_COST)
st.debug_run(A.RUN_AGGREGATE, 101)
b = st.debug_read(A.BUF_VOLUME_A).copy()
bad = (a != b)
print("shape", a.shape, "bad", bad.sum())
ys, xs, ds = np.nonzero(bad)
print("bad x histogram (min,max):", xs.min() if len(xs) else None, xs.max() if len(xs) else None)
print("bad per x (first 20 distinct):", np.unique(xs)[:40])
print("rows with bad:", len(np.unique(ys)), "of", h)
y0 = ys[0]; print("row", y0, "bad x:", np.unique(xs[ys == y0])[:30], "d:", np.unique(ds[ys == y0])[:20])
print("row y0 x=436..449 d=0 a:", a[y0, 436:450, 0].tolist()); print("b:", b[y0, 436:450, 0].tolist())
print("arms of row", y0, "x=375..390:", st.debug_read(A.BUF_ARMS)[y0, 375:392].tolist())
PY
