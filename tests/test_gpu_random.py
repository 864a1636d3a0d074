"""GPU tier: randomly drawn geometries and OPTIONS against the CPU oracle -- aimed at the region voting of round 5 (all ten passes
of multistep_refiner.cpp:153-227 as one fixed-point iteration, irv_plan.h), whose state encoding, band / workgroup layout and vote
depend on image size, disparity range, arm limits and the two voting thresholds: the voting stage in isolation (oracle's LR-checked
map, labels and arms in; oracle's map after the voting out) and the whole Match, bit for bit."""
import numpy as np
import pytest

from adcensus_amd import workloads
from oracle import pyoracle
from tests import cases

pytestmark = pytest.mark.gpu


def _draw(rng):
    w, h = int(rng.integers(40, 520)), int(rng.integers(24, 300))
    d = int(rng.choice([16, 37, 64, 100, 128, 192]))
    dmin = int(rng.choice([0, 0, 0, -9, 5]))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        pair = workloads.structured_pair(w, h, d, seed=int(rng.integers(1, 1 << 30)))
    elif kind == 1:
        pair = workloads.quantized_noise_pair(w, h, d, seed=int(rng.integers(1, 1 << 30)), levels=int(rng.choice([16, 32, 64])))
    else:
        pair = workloads.noise_pair(w, h, seed=int(rng.integers(1, 1 << 30)))
    l1 = int(rng.choice([4, 9, 17, 34, 34, 50]))
    opt = pyoracle.Option(min_disparity=dmin, max_disparity=dmin + d, cross_L1=l1, cross_L2=max(1, l1 // 2),
                          cross_t1=int(rng.integers(8, 40)), cross_t2=int(rng.integers(3, 12)),
                          irv_ts=int(rng.choice([0, 5, 20, 20, 45])), irv_th=float(rng.choice([0.1, 0.3, 0.4, 0.4, 0.7])),
                          lrcheck_thres=float(rng.choice([0.5, 1.0, 1.0, 2.0])))
    return pair, opt


@pytest.mark.parametrize("seed", [101, 202, 303, 404])
def test_random_geometries_and_options_equal_oracle(hip, oracle, seed):
    A = hip
    rng = np.random.default_rng(seed)
    bad = []
    for k in range(6):
        (left, right), opt = _draw(rng)
        h, w = left.shape[:2]
        o = oracle.run(left, right, opt)
        tag = "seed %d case %d: %dx%d [%d, %d) L1 %d ts %d th %.1f" % (seed, k, w, h, opt.min_disparity, opt.max_disparity, opt.cross_L1,
                                                                      opt.irv_ts, opt.irv_th)
        st = A.ADCensusStereo(device=0)
        assert st.Initialize(w, h, cases.to_product_option(opt)), tag
        # the voting stage in isolation
        st.debug_set_images(left, right)
        st.debug_write(A.BUF_ARMS, o["arms"])
        st.debug_write(A.BUF_SUPCOUNT_H, o["sup_count_h"])
        st.debug_write(A.BUF_DISP_LEFT, o["disp_after_lr"])
        st.debug_write(A.BUF_OUTLIER_LABEL, o["outlier_label"])
        st.debug_run(A.RUN_REGION_VOTING)
        got = st.debug_read(A.BUF_DISP_LEFT)
        if not np.array_equal(np.asarray(got).view(np.uint32), o["disp_after_irv"].view(np.uint32)):
            bad.append(tag + " (voting stage: %d pixels)" % int((np.asarray(got).view(np.uint32) != o["disp_after_irv"].view(np.uint32)).sum()))
        # the whole Match, twice (the second one runs on the adapted voting budget / assumed ring depth)
        for rep in range(2):
            d = st.match(left, right)
            if not np.array_equal(d.view(np.uint32), o["disp_final"].view(np.uint32)):
                bad.append(tag + " (Match %d: %d pixels)" % (rep, int((d.view(np.uint32) != o["disp_final"].view(np.uint32)).sum())))
        st.Release()
    assert not bad, bad
