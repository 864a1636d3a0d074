"""GPU tier at BASELINE.json's FULL size (1920x1080, D=128): the whole Match against the reference CPU program
(~20 s of host time per pair), repeatability, and the equality of the aggregation paths (plain 8 passes vs fused
cost + pass pairs) through the stage-level debug surface.  Exercises what the small cases cannot: many segments per
line, all 17 median bands, a long voting chain, multi-round grids."""
import numpy as np
import pytest

from adcensus_amd import workloads
from oracle import pyoracle
from tests import cases

pytestmark = pytest.mark.gpu

W, H, D = 1920, 1080, 128


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


@pytest.mark.parametrize("workload,seed", [("noise", 12345), ("structured", 777), ("noise", 12346), ("structured", 778)])
def test_full_size_match_equals_reference(hip, oracle, workload, seed):
    """Pairs 0 and 1 of bench.py's batches (noise: seeds 12345 + i, structured: 777 + i) against the reference CPU program."""
    A = hip
    left, right = (workloads.noise_pair(W, H, seed) if workload == "noise"
                   else workloads.structured_pair(W, H, D, seed=seed))
    opt = pyoracle.Option(max_disparity=D)
    want, _ = oracle.match(left, right, opt)
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(W, H, cases.to_product_option(opt))
    got = np.zeros((H, W), np.float32)
    assert st.Match(left, right, got)
    bad = int((got.view(np.uint32) != want.view(np.uint32)).sum())
    assert bad == 0, "%s: %d of %d pixels differ from the reference" % (workload, bad, W * H)
    again = np.zeros((H, W), np.float32)
    assert st.Match(left, right, again) and _same(again, got)  # repeatable (no order dependence between waves)
    st.Release()


def _ref_table():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "farm_ref_digests.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("workload", ["noise", "structured"])
def test_bench_batches_equal_reference_digests(hip, workload):
    """EVERY pair of the two batches bench.py measures on (noise seeds 12345 + i, i < 20 = the headline batch; structured
    seeds 777 + i, i < 10) against the SHA-256 of the reference CPU program's map for that pair
    (tests/golden/farm_ref_digests.json, made by tools/make_farm_ref_digests.py from oracle/_ref: ~55 s of one core per pair,
    so the table is committed instead of recomputed here).  One handle takes the whole batch in order, like the farm does: the
    speculative forms (scanline row segments, median bands, assumed ring depth / voting budget) see 20 different images."""
    import hashlib
    A = hip
    t = _ref_table()
    assert t["size"] == [W, H, D] and len(t["noise"]) >= 20 and len(t["structured"]) >= 10
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(W, H, A.ADCensusOption(max_disparity=D))
    got = np.zeros((H, W), np.float32)
    bad = []
    for pid in sorted(int(k) for k in t[workload]):
        left, right = (workloads.noise_pair(W, H, 12345 + pid) if workload == "noise"
                       else workloads.structured_pair(W, H, D, seed=777 + pid))
        got[:] = -1.0
        assert st.Match(left, right, got)
        if hashlib.sha256(got.tobytes()).hexdigest() != t[workload][str(pid)]:
            bad.append(pid)
    st.Release()
    assert not bad, "%s pairs %s differ from the reference CPU program's maps" % (workload, bad)


@pytest.mark.parametrize("workload", ["noise", "structured"])
def test_kitti_size_match_equals_reference(hip, oracle, workload):
    """BASELINE.json configs[2]: KITTI-size 1242x375, D=128 (odd width, 375 rows = 5.9 median bands, 375 scanline rows)."""
    A = hip
    w, h, d = 1242, 375, 128
    left, right = (workloads.noise_pair(w, h, 4242) if workload == "noise" else workloads.structured_pair(w, h, d, seed=4243))
    opt = pyoracle.Option(max_disparity=d)
    want, _ = oracle.match(left, right, opt)
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(w, h, cases.to_product_option(opt))
    got = np.zeros((h, w), np.float32)
    assert st.Match(left, right, got)
    bad = int((got.view(np.uint32) != want.view(np.uint32)).sum())
    assert bad == 0, "%s: %d of %d pixels differ from the reference" % (workload, bad, w * h)
    st.Release()


def test_full_size_aggregation_paths_agree(hip):
    """plain 8-pass aggregation == fused cost + host-chosen ring + pass pairs, bit for bit, on the noise pair
    (short arms: small ring, pairs) and on the structured pair (long arms: full ring)."""
    A = hip
    for left, right in (workloads.noise_pair(W, H, 12345), workloads.structured_pair(W, H, D, seed=777)):
        st = A.ADCensusStereo(device=0)
        assert st.Initialize(W, H, A.ADCensusOption(max_disparity=D))
        st.debug_set_images(left, right)
        st.debug_run(A.RUN_GRAY_CENSUS)
        st.debug_run(A.RUN_COST)
        st.debug_run(A.RUN_ARMS)
        st.debug_run(A.RUN_AGGREGATE, 4)
        plain = st.debug_read(A.BUF_VOLUME_A).copy()
        st.debug_run(A.RUN_COST)  # (overwritten on purpose: the fused variant must not read it)
        st.debug_run(A.RUN_AGGREGATE, 304)
        assert _same(st.debug_read(A.BUF_VOLUME_A), plain)
        st.Release()


def test_full_size_stage_isolation_structured(hip, oracle):
    """Every stage in ISOLATION at 1920x1080, D=128 on a structured pair (seed 779: long arms -> the pair-register-ring
    aggregation with balanced chunks, ~350 voting kernels, 17 median bands): the reference's dump of stage k-1 in, HIP stage k,
    compared with the reference's dump of stage k -- a full-size regression names the stage that broke (the volumes are
    1.06 GB each; they travel through the debug surface)."""
    from tests import gpu_harness
    left, right = workloads.structured_pair(W, H, D, seed=779)
    opt = pyoracle.Option(max_disparity=D)
    o = oracle.run(left, right, opt)
    rep = gpu_harness.stage_report(left, right, opt, o)
    bad = gpu_harness.failing(rep)
    assert not bad, "structured 1080p (oracle=%s): %s" % (oracle.kind, bad)
    # the chain really ran a long fixed-point iteration (round 5: all ten passes iterate at once -- ~80 rounds where the
    # pass-after-pass chain of rounds 1-4 took ~255), and every entry was evaluated at least once
    assert rep["disp_after_irv"]["voting_rounds_evals"][0] > 30 and rep["disp_after_irv"]["voting_rounds_evals"][1] > 100000
