"""Region-voting chain on one structured case against the oracle, for three launch budgets (the smallest forces the host-side
continuation).  The launch shape comes from the environment (ADC_IRV_GRID workgroups x ADC_IRV_WPB waves): small shapes make
the work list span several batches (tests/test_gpu_api.py runs this script in a subprocess because the shape is read once
per process).  Exit code 1 on any difference."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import adcensus_amd as A
from adcensus_amd import workloads
from oracle import pyoracle
from tests import cases
w, h, d = 200, 150, int(os.environ.get("ADC_CHECK_D", "32"))  # (an odd range: the LDS pool behind the histograms must stay aligned)
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
    worst = max(globals().get("worst", 0), bad + abs(filled_ref - filled_got))
    print("budget", budget, "bad", bad, "filled ref/got", filled_ref, filled_got, "stats(rounds,evals)", st.voting_stats(), "overflows", st.debug_counter(1), "next budget", st.debug_counter(3), flush=True)
    st.Release()
sys.exit(1 if worst else 0)
