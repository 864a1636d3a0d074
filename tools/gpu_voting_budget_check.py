import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import adcensus_amd as A
from adcensus_amd import workloads
from oracle import pyoracle
from tests import cases
w, h, d = 200, 150, 32
left, right = workloads.structured_pair(w, h, d, seed=5)
opt = pyoracle.Option(max_disparity=d)
o = pyoracle.load("auto").run(left, right, opt)
for budget in (800, 96, 4):
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(w, h, cases.to_product_option(opt))
    st.debug_set_images(left, right)
    st.debug_run(A.RUN_ARMS)
    st.debug_write(A.BUF_ARMS, o["arms"]); st.debug_write(A.BUF_SUPCOUNT_H, o["sup_count_h"]); st.debug_write(A.BUF_SUPCOUNT_V, o["sup_count_v"])
    st.debug_write(A.BUF_DISP_LEFT, o["disp_after_lr"]); st.debug_write(A.BUF_OUTLIER_LABEL, o["outlier_label"])
    st.debug_run(A.RUN_REGION_VOTING, budget)
    got = st.debug_read(A.BUF_DISP_LEFT)
    bad = int((got.view(np.uint32) != o["disp_after_irv"].view(np.uint32)).sum())
    filled_ref = int((np.isinf(o["disp_after_lr"]) & ~np.isinf(o["disp_after_irv"])).sum())
    filled_got = int((np.isinf(o["disp_after_lr"]) & ~np.isinf(got)).sum())
    print("budget", budget, "bad", bad, "filled ref/got", filled_ref, filled_got, "stats(rounds,evals)", st.voting_stats(), "overflows", st.debug_counter(1), "next budget", st.debug_counter(3), flush=True)
    st.Release()
