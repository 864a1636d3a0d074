"""CPU tier: the order-aware PARALLEL formulations used by the HIP kernels (marching-ring sums,
closed-form sticky d2, two-phase LR check, fixed-point region voting with dirty tiles and arbitrary
evaluation order, Jacobi interpolation, level-synchronous median) are emulated lane-by-lane on the
CPU (tests/emul/emul.cpp, sharing adc_device_fn.h with the device code) and must equal the oracle's
sequential in-place results bit-for-bit."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


EMUL_CASES = ["cone_crop_d40", "s2_96x64_d32", "q_20x40_d32", "q_9x20_d8", "q_30x7_d8", "q_1x40_d8", "q_40x1_d8",
              "q_3x3_d2", "s2_150x100_neg", "q_40x30_pos_wltd", "s2_150x100_pos"]
# + disparity ranges that put 2 and 4 disparities into a lane (scanline class derivation, winner-takes-all)
LANE_CASES = EMUL_CASES + ["s2_320x180_d128", "s2_200x120_d200", "noise_160x90_d128_pos", "s2_200x120_d160",
                           "noise_96x50_d160_neg", "s2_360x60_d300", "noise_80x40_d520"]


@pytest.fixture(scope="module")
def dumps(port_oracle):
    cache = {}

    def get(name):
        if name not in cache:
            left, right, opt = cases.make_case(name)
            cache[name] = (left, right, opt, port_oracle.run(left, right, opt))
        return cache[name]
    return get


@pytest.mark.parametrize("name", ["s2_96x64_d32", "q_20x40_d32", "q_9x20_d8", "q_1x40_d8", "q_40x1_d8", "q_3x3_d2"])
@pytest.mark.parametrize("pf,hseg,vseg", [(8, 1, 1), (8, 4, 2), (5, 3, 1), (2, 1, 7)])
def test_marching_ring_aggregation(emul, dumps, name, pf, hseg, vseg):
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D = opt.max_disparity - opt.min_disparity
    a, b = o["cost_init"].copy(), np.empty_like(o["cost_init"])
    L = max(0, min(opt.cross_L1, 255))
    hfirst = True
    for _ in range(4):
        order = [(0, 0, o["sup_count_h"]), (1, 1, o["sup_count_h"])] if hfirst else [(1, 0, o["sup_count_v"]), (0, 1, o["sup_count_v"])]
        for vert, div, sup in order:
            emul.emul_aggregate_pass(P(a), P(b), P(o["arms"]), P(sup), w, h, D, vert, div, L, vseg if vert else hseg, pf)
            a, b = b, a
        hfirst = not hfirst
    assert same(a, o["cost_aggr"])


@pytest.mark.parametrize("name", ["s2_96x64_d32", "q_20x40_d32", "q_9x20_d8", "q_1x40_d8", "q_40x1_d8", "q_3x3_d2"])
@pytest.mark.parametrize("pf,hseg,vseg", [(8, 1, 1), (8, 3, 2), (3, 2, 5)])
def test_marching_ring_pass_pairs(emul, dumps, name, pf, hseg, vseg):
    """Production launch sequence on short-arm images: H0 | V0+V1 | H1+H2 | V2+V3 | H3 (8 passes in 5 launches); a pair =
    dividing pass + the next iteration's first pass through a second ring, segment halos of 2L."""
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D = opt.max_disparity - opt.min_disparity
    L = max(0, min(opt.cross_L1, 255))
    a, b = o["cost_init"].copy(), np.empty_like(o["cost_init"])
    emul.emul_aggregate_pass(P(a), P(b), P(o["arms"]), P(o["sup_count_h"]), w, h, D, 0, 0, L, hseg, pf)  # H0
    a, b = b, a
    for vert, sup in ((1, o["sup_count_h"]), (0, o["sup_count_v"]), (1, o["sup_count_h"])):  # the three pairs
        emul.emul_aggregate_pair(P(a), P(b), P(o["arms"]), P(sup), w, h, D, vert, L, vseg if vert else hseg, pf)
        a, b = b, a
    emul.emul_aggregate_pass(P(a), P(b), P(o["arms"]), P(o["sup_count_v"]), w, h, D, 0, 1, L, hseg, pf)  # H3 (dividing)
    a, b = b, a
    assert same(a, o["cost_aggr"])


RR_CASES = ["s2_96x64_d32", "q_20x40_d32", "q_9x20_d8", "q_1x40_d8", "q_40x1_d8", "q_3x3_d2", "cone_crop_d40", "q_257x131_d64"]


@pytest.mark.parametrize("name", RR_CASES)
@pytest.mark.parametrize("hseg,vseg", [(1, 1), (3, 2), (2, 5)])
def test_register_ring_body_single_passes(emul, dumps, name, hseg, vseg):
    """k_aggregate_rr.h compiled for the CPU (RR_EMUL): 8 single register-ring passes == the reference's cost_aggr
    (biased-arm records, one slot counter per ring, 35-add blocks with M0-relative addressing, Markstein division)."""
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D = opt.max_disparity - opt.min_disparity
    L = max(0, min(opt.cross_L1, 255))
    a, b = o["cost_init"].copy(), np.full_like(o["cost_init"], np.nan)
    hfirst = True
    for _ in range(4):
        order = [(0, 0, o["sup_count_h"]), (1, 1, o["sup_count_h"])] if hfirst else [(1, 0, o["sup_count_v"]), (0, 1, o["sup_count_v"])]
        for vert, div, sup in order:
            assert emul.emul_rr_pass(P(a), P(b), P(o["arms"]), P(sup), w, h, D, vert, div, 0, L, vseg if vert else hseg) == 0
            a, b = b, a
        hfirst = not hfirst
    assert same(a, o["cost_aggr"])


@pytest.mark.parametrize("name", RR_CASES)
@pytest.mark.parametrize("hseg,vseg", [(1, 1), (3, 2), (2, 5)])
def test_register_ring_body_pass_pairs(emul, dumps, name, hseg, vseg):
    """Production launch sequence on long-arm images: H0 | V0+V1 | H1+H2 | V2+V3 | H3 with the pairs on two register
    rings (k_agg_regring_pair): first-pass outputs stay in ring 2, halo steps store to the sink."""
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D = opt.max_disparity - opt.min_disparity
    L = max(0, min(opt.cross_L1, 255))
    a, b = o["cost_init"].copy(), np.full_like(o["cost_init"], np.nan)
    assert emul.emul_rr_pass(P(a), P(b), P(o["arms"]), P(o["sup_count_h"]), w, h, D, 0, 0, 0, L, hseg) == 0  # H0
    a, b = b, a
    for vert, sup in ((1, o["sup_count_h"]), (0, o["sup_count_v"]), (1, o["sup_count_h"])):  # the three pairs
        assert emul.emul_rr_pass(P(a), P(b), P(o["arms"]), P(sup), w, h, D, vert, 1, 1, L, vseg if vert else hseg) == 0
        a, b = b, a
    assert emul.emul_rr_pass(P(a), P(b), P(o["arms"]), P(o["sup_count_v"]), w, h, D, 0, 1, 0, L, hseg) == 0  # H3 (dividing)
    a, b = b, a
    assert same(a, o["cost_aggr"])


RR2_CASES = RR_CASES + ["s2_320x180_d128", "s2_200x120_d200", "noise_160x90_d128_pos", "s2_150x100_neg"]


@pytest.mark.parametrize("name", RR2_CASES)
@pytest.mark.parametrize("hchunk,vchunk,fused", [(0, 0, True), (37, 23, True), (1000, 7, False), (13, 250, True)])
def test_register_ring_pairs_body(emul, dumps, name, hchunk, vchunk, fused):
    """k_aggregate_rr2.h compiled for the CPU (RR_EMUL): two disparities per lane, ring slots = VGPR pairs addressed with
    M0 = 2 * slot, packed 35-add blocks; 8 single passes == the reference's cost_aggr.  A wave = one chunk of the line-major
    output index space (0 = whole lines; chunks that straddle line ends, cover several lines, or are shorter than the
    prefetch depth).  fused: the first pass computes the matching cost itself from packed pixel records (two lane
    windows) instead of reading cost_init."""
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D, dmin = opt.max_disparity - opt.min_disparity, opt.min_disparity
    L = max(0, min(opt.cross_L1, 255))
    a, b = o["cost_init"].copy(), np.full_like(o["cost_init"], np.nan)
    hfirst, first = True, True
    for _ in range(4):
        order = [(0, 0, o["sup_count_h"]), (1, 1, o["sup_count_h"])] if hfirst else [(1, 0, o["sup_count_v"]), (0, 1, o["sup_count_v"])]
        for vert, div, sup in order:
            ci = 1 if (fused and first) else 0
            if ci:
                a[:] = np.nan  # the fused pass must not read the cost volume
            rc = emul.emul_rr2_pass(P(a), P(b), P(o["arms"]), P(sup), w, h, D, vert, div, L, vchunk if vert else hchunk, ci,
                                    P(left), P(right), P(o["census_left"]), P(o["census_right"]), dmin, opt.lambda_ad, opt.lambda_census)
            assert rc == 0
            first = False
            a, b = b, a
        hfirst = not hfirst
    assert same(a, o["cost_aggr"])


def test_rr2_cost_windows_shift_identity(emul):
    """The two lane windows of the fused cost at two disparities per lane: A' = shr(B) with the new column at lane 0,
    B' = A keeps lane l on the columns x - 2l and x - 2l - 1 (whole-wave model of RR2_WIN_STEP)."""
    for start in (0, 5, 1000):
        assert emul.emul_rr2_window_identity(300, start) == 0


@pytest.mark.parametrize("name", ["s2_96x64_d32", "q_20x40_d32", "q_9x20_d8", "q_1x40_d8", "q_40x1_d8", "q_3x3_d2", "s2_150x100_neg",
                                  "s2_200x120_d200", "q_40x30_pos_wltd", "s2_150x100_pos", "s2_200x120_d160", "noise_96x50_d160_neg"])
@pytest.mark.parametrize("seg", [0, 7, 50])
def test_fused_cost_lane_window(emul, dumps, name, seg):
    """The matching cost as the fused first aggregation pass computes it (shifting lane window over padded right-image
    records, packed-colour SAD, census popcount, host tables) == the reference's cost volume, bit for bit."""
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D, dmin = opt.max_disparity - opt.min_disparity, opt.min_disparity
    got = np.full((h, w, D), np.nan, np.float32)
    emul.emul_cost_window(P(left), P(right), P(o["census_left"]), P(o["census_right"]), P(got), w, h, dmin, D,
                          opt.lambda_ad, opt.lambda_census, seg)
    assert same(got, o["cost_init"])


def test_register_ring_span_addressing(emul):
    """agg_reg_sum: one M0 value per block of 16 adds with static register numbers v40..v55, entered late -- every read
    lands on ring[idx .. idx+cnt) (registers v56+) in increasing order, for every start slot and span length."""
    import ctypes as C
    emul.emul_regring_span.restype = C.c_float
    emul.emul_regring_span.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rng = np.random.default_rng(3)
    R = 69
    vgpr = np.full(256, np.nan, np.float32)  # anything outside the ring poisons the sum
    ring = (rng.random(R, dtype=np.float32) * 7 + 0.1).astype(np.float32)
    vgpr[56:56 + R] = ring
    lo, hi = C.c_int(), C.c_int()
    for idx in range(R):
        for cnt in range(1, R - idx + 1):  # a run never wraps (the wrap is a second run)
            got = emul.emul_regring_span(vgpr.ctypes.data, idx, cnt, C.byref(lo), C.byref(hi))
            want = np.float32(0.0)
            for v in ring[idx:idx + cnt]:
                want = np.float32(want + v)
            assert np.float32(got).view(np.uint32) == want.view(np.uint32), (idx, cnt)
            assert lo.value == 56 + idx and hi.value == 56 + idx + cnt - 1 and hi.value < 128


@pytest.mark.parametrize("name", LANE_CASES)
def test_scanline_closed_form(emul, dumps, name):
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D, dmin = opt.max_disparity - opt.min_disparity, opt.min_disparity
    lh, lv, rh, rv = (np.zeros((h, w), np.uint8) for _ in range(4))
    emul.emul_color_diffs(P(left), P(lh), P(lv), w, h)
    emul.emul_color_diffs(P(right), P(rh), P(rv), w, h)
    a, b = o["cost_aggr"].copy(), np.empty_like(o["cost_aggr"])
    for vert, dr in ((0, 1), (0, -1), (1, 1), (1, -1)):
        emul.emul_scanline_pass(P(a), P(b), P(lv if vert else lh), P(rv if vert else rh), w, h, dmin, D, vert, dr,
                                opt.so_tso, C.c_float(opt.so_p1), C.c_float(opt.so_p2))
        a, b = b, a
    assert same(a, o["cost_so"])
    # the kernel's per-lane class derivation (bytes of the right-image step map + closed-form sticky-d2 rule)
    a, b = o["cost_aggr"].copy(), np.empty_like(o["cost_aggr"])
    for vert, dr in ((0, 1), (0, -1), (1, 1), (1, -1)):
        emul.emul_scanline_pass_lanes(P(a), P(b), P(lv if vert else lh), P(rv if vert else rh), w, h, dmin, D, vert, dr,
                                      opt.so_tso, C.c_float(opt.so_p1), C.c_float(opt.so_p2))
        a, b = b, a
    assert same(a, o["cost_so"])


@pytest.mark.parametrize("name,want_short", [("s2_320x180_d128", True), ("q_257x131_d64", True), ("cone_neg", True), ("cone_pos", True),
                                             ("noise_160x90_d128_pos", True), ("s2_150x100_neg", False), ("q_30x7_d8", False)])
def test_scanline_chunked_forms(emul, dumps, name, want_short):
    """Control flow of the asm-prefetch scanline kernels: 16-step chunks, the short form of whole interior chunks
    (adc_so_chunk_interior: interior class rule without its test, rmap offsets of the prefetched elements = one clamped offset
    + a running step), the clamped general form everywhere else, and the transitions between the two -- against the
    reference's scanline result.  want_short: the case is wide enough (and D = 64 * VPL) for the short form to occur."""
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D, dmin = opt.max_disparity - opt.min_disparity, opt.min_disparity
    lh, lv, rh, rv = (np.zeros((h, w), np.uint8) for _ in range(4))
    emul.emul_color_diffs(P(left), P(lh), P(lv), w, h)
    emul.emul_color_diffs(P(right), P(rh), P(rv), w, h)
    a, b = o["cost_aggr"].copy(), np.empty_like(o["cost_aggr"])
    short = 0
    for vert, dr in ((0, 1), (0, -1), (1, 1), (1, -1)):
        n = emul.emul_scanline_pass_chunked(P(a), P(b), P(lv if vert else lh), P(rv if vert else rh), w, h, dmin, D, vert, dr,
                                            opt.so_tso, C.c_float(opt.so_p1), C.c_float(opt.so_p2))
        assert n >= 0
        short += n
        a, b = b, a
    assert same(a, o["cost_so"])
    assert (short > 0) == want_short, short


@pytest.mark.parametrize("name", ["s2_320x180_d128", "q_257x131_d64", "cone_crop_d40", "s2_150x100_neg", "noise_160x90_d128_pos"])
def test_scanline_verified_segments(emul, dumps, name):
    """Paths cut into segments that start with the wrong state 64 elements early and are verified bit for bit at the seam
    (DESIGN 4.2): the four chained passes still give the reference's volume; with a 64-element warm-up no seam fails on these
    cases, with a 4-element warm-up most do -- and the result is still exact, because a failed seam falls back to the
    predecessor's state."""
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D, dmin = opt.max_disparity - opt.min_disparity, opt.min_disparity
    lh, lv, rh, rv = (np.zeros((h, w), np.uint8) for _ in range(4))
    emul.emul_color_diffs(P(left), P(lh), P(lv), w, h)
    emul.emul_color_diffs(P(right), P(rh), P(rv), w, h)
    emul.emul_scanline_pass_segments.restype = C.c_long
    for nseg, warm, expect_clean in ((2, 64, True), (3, 64, True), (3, 4, False)):
        a, b = o["cost_aggr"].copy(), np.empty_like(o["cost_aggr"])
        failed = 0
        for vert, dr in ((0, 1), (0, -1), (1, 1), (1, -1)):
            failed += emul.emul_scanline_pass_segments(P(a), P(b), P(lv if vert else lh), P(rv if vert else rh), w, h, dmin, D, vert, dr,
                                                       opt.so_tso, C.c_float(opt.so_p1), C.c_float(opt.so_p2), nseg, warm)
            a, b = b, a
        assert same(a, o["cost_so"]), (nseg, warm)
        if expect_clean:
            assert failed == 0, (nseg, warm, failed)
        elif min(w, h) >= 96:
            assert failed > 0


@pytest.mark.parametrize("name", ["cone_crop_d40", "s2_150x100_neg", "s2_150x100_pos", "s2_320x180_d128", "noise_160x90_d128_pos"])
def test_scanline_kernel_segments(emul, dumps, name):
    """The row passes in the form k_scanline_seg runs them (round 4): segment bounds from adc_so_seg_start (first outputs = 1 mod 4,
    warm + 1 elements of overlap, equal step counts), warm-up outputs in the seam slot, seam check behind the pass.  With the
    production warm-up (64) no seam fails and the two row passes equal whole-row passes bit for bit; with a 16-step warm-up seams
    fail on the larger cases -- which is all the product needs to know (it redoes the Match with whole rows)."""
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D, dmin = opt.max_disparity - opt.min_disparity, opt.min_disparity
    lh, lv, rh, rv = (np.zeros((h, w), np.uint8) for _ in range(4))
    emul.emul_color_diffs(P(left), P(lh), P(lv), w, h)
    emul.emul_color_diffs(P(right), P(rh), P(rv), w, h)
    emul.emul_scanline_pass_kernel_segments.restype = C.c_long
    emul.emul_scanline_pass_segments.restype = C.c_long
    emul.emul_so_seg_ok.restype = C.c_int
    # whole-row passes (nseg = 1 of the older model = the plain chained passes)
    ref_a, ref_b = o["cost_aggr"].copy(), np.empty_like(o["cost_aggr"])
    for dr in (1, -1):
        emul.emul_scanline_pass_segments(P(ref_a), P(ref_b), P(lh), P(rh), w, h, dmin, D, 0, dr, opt.so_tso, C.c_float(opt.so_p1),
                                         C.c_float(opt.so_p2), 1, 64)
        ref_a, ref_b = ref_b, ref_a
    ran = 0
    for nseg, warm in ((2, 64), (3, 64), (5, 64), (2, 16), (4, 16)):
        if not emul.emul_so_seg_ok(w, nseg, warm):
            continue
        ran += 1
        a, b = o["cost_aggr"].copy(), np.empty_like(o["cost_aggr"])
        failed = 0
        for dr in (1, -1):
            r = emul.emul_scanline_pass_kernel_segments(P(a), P(b), P(lh), P(rh), w, h, dmin, D, 0, dr, opt.so_tso, C.c_float(opt.so_p1),
                                                        C.c_float(opt.so_p2), nseg, warm)
            assert r >= 0, (nseg, warm, r)
            failed += r
            a, b = b, a
        if warm == 64:
            assert failed == 0, (nseg, warm, failed)
        if failed == 0:
            assert same(a, ref_a), (nseg, warm)
    assert ran >= 1


def test_scanline_segment_bounds(emul):
    """adc_so_seg_start / adc_so_seg_ok: first outputs = 1 (mod 4) (the kernel's d1 word groups), strictly increasing, the
    warm-up fits in front of every later segment, every segment keeps two chunks of outputs, step counts within 8 of each other."""
    emul.emul_so_seg_ok.restype = C.c_int
    emul.emul_so_seg_start.restype = C.c_int
    checked = 0
    for plen in list(range(40, 700, 7)) + [1242, 1920, 2048, 3840]:
        for warm in (16, 64):
            for nseg in range(2, 9):
                if not emul.emul_so_seg_ok(plen, nseg, warm):
                    continue
                checked += 1
                st = [emul.emul_so_seg_start(plen, nseg, warm, s) for s in range(nseg + 1)]
                assert st[0] == 0 and st[-1] == plen
                steps = []
                for s in range(nseg):
                    a, b = st[s], st[s + 1]
                    assert b - a >= 32
                    if s > 0:
                        assert a % 4 == 1 and a - warm - 1 >= 0 and (a - warm - 1) % 4 == 0
                    steps.append(b - (a - warm - 1 if s else 0))
                assert max(steps[:-1] + [steps[-1]]) - min(steps[:-1]) <= 8 or nseg == 2, (plen, nseg, warm, steps)
    assert checked > 500
    assert emul.emul_so_seg_ok(1920, 2, 64) and emul.emul_so_seg_ok(1242, 5, 64) and not emul.emul_so_seg_ok(1920, 2, 40)


def test_scanline_chunk_predicate_implies_no_clamp_and_interior_rule(emul):
    """Exhaustive over small geometries: whenever adc_so_chunk_interior accepts a chunk, (i) every element it steps on is
    interior in the sense of the per-step test, (ii) the clamped rmap offset of every element it prefetches is affine in the
    element index for every lane (so the running offset of the short form is exact), (iii) the d1 word groups it prefetches
    exist.  (The kernel's own expressions are the ones restated in adc_device_fn.h: checked textually below.)"""
    bad = emul.emul_so_chunk_predicate_check()
    assert bad == 0, bad
    src = open(os.path.join(ROOT, "adcensus_amd", "csrc", "k_scanline.hip")).read()
    for expr in ("xlo >= dmin + Dp && xhi - dmin < W - 1", "i + 2 * PF + 4 <= plen_v", "const int ea = e0 + i, eb = e0 + i + 2 * PF - 1;",
                 "return sy * g.W + (xr > 1 ? xr : 1) + ((!VERT && g.dir < 0) ? 1 : 0);", "xr = xr > g.W - 1 ? g.W - 1 : xr;"):
        assert expr in src, "k_scanline.hip no longer contains `%s`: update adc_device_fn.h's restatement with it" % expr


@pytest.mark.parametrize("name", LANE_CASES)
def test_wta(emul, dumps, name):
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D, dmin = opt.max_disparity - opt.min_disparity, opt.min_disparity
    dl, dr = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
    emul.emul_wta(P(o["cost_so"]), P(dl), w, h, dmin, D, 0)
    emul.emul_wta(P(o["cost_so"]), P(dr), w, h, dmin, D, 1)
    assert same(dl, o["disp_left_wta"])
    assert same(dr, o["disp_right_wta"])
    # diagonal-band form of the right view (k_wta_right_band): per-lane sequential scan
    db = np.empty((h, w), np.float32)
    emul.emul_wta_right_band(P(o["cost_so"]), P(db), w, h, dmin, D)
    mask = np.ones((h, w), bool)
    if dmin > 0:
        mask[:, w - dmin:] = False  # reference reads out of bounds there (documented, tests/gpu_harness.py)
    assert same(db[mask], o["disp_right_wta"][mask])
    # marching form (k_wta_right_march, D <= 128): whole rows, and the plan's segments for several "CU counts"
    if D <= 128:
        emul.emul_wta_right_march.restype = C.c_long
        for ncu, nseg in ((256, 0), (7, 0), (3, 0), (1, 0), (h + 1, 2), (h + 1, 3), (5, 8)):
            dm = np.full((h, w), -7.0, np.float32)
            units = emul.emul_wta_right_march(P(o["cost_so"]), P(dm), w, h, dmin, D, ncu, nseg)
            assert units >= h, (ncu, nseg, units)
            if nseg and ncu > h:
                assert units == h * min(nseg, (w + 63) // 64), (ncu, nseg, units)
            assert same(dm[mask], o["disp_right_wta"][mask]), (ncu, nseg)


@pytest.mark.parametrize("name", EMUL_CASES)
def test_refiner_parallel_forms(emul, dumps, name):
    left, right, opt, o = dumps(name)
    h, w = left.shape[:2]
    D, dmin = opt.max_disparity - opt.min_disparity, opt.min_disparity
    out, lab = np.empty((h, w), np.float32), np.empty((h, w), np.uint8)
    emul.emul_lrcheck(P(o["disp_left_wta"]), P(o["disp_right_wta"]), P(out), P(lab), w, h, C.c_float(opt.lrcheck_thres))
    assert same(out, o["disp_after_lr"]) and same(lab, o["outlier_label"])
    # fixed-point voting, two different (shuffled) evaluation orders
    emul.emul_region_voting.restype = C.c_long
    for seed in (1, 99):
        d = o["disp_after_lr"].copy()
        ev = C.c_long(0)
        emul.emul_region_voting(P(d), P(o["outlier_label"]), P(o["arms"]), w, h, dmin, D, opt.irv_ts, C.c_float(opt.irv_th),
                                max(0, min(opt.cross_L1, 255)), seed, C.byref(ev))
        assert same(d, o["disp_after_irv"])
    # the device-driven chain (k_voting.hip): state machine of irv_plan.h on the 16-bit state map, shuffled vote order
    emul.emul_irv_chain.restype = C.c_long
    # (the grid decides the work-list layout; 2 x 4 and 5 x 1 waves = batches of 512 / 320 entries: lists span several batches)
    # 16 x 1: a multiple of 8 workgroups -> whole bands belong to one "XCD" (irv_wg_tile)
    for seed, groups, wpb in ((3, 2, 4), (77, 5, 1), (9, 16, 1)):
        d = o["disp_after_lr"].copy()
        stats = (C.c_long * 3)()
        L = max(0, min(opt.cross_L1, 255))
        rounds = emul.emul_irv_chain(P(d), P(o["outlier_label"]), P(o["arms"]), P(o["sup_count_h"]), w, h, dmin, D, opt.irv_ts,
                                     C.c_float(opt.irv_th), opt.irv_ts if L <= 127 else -1, seed, groups, wpb, stats)
        assert rounds >= 0, rounds
        assert same(d, o["disp_after_irv"])
        # the same chain without the slack budgets of round 6 (ADC_IRV_SLACK=0): same result, never fewer evaluations
        emul.emul_irv_chain2.restype = C.c_long
        d0, stats0 = o["disp_after_lr"].copy(), (C.c_long * 3)()
        assert emul.emul_irv_chain2(P(d0), P(o["outlier_label"]), P(o["arms"]), P(o["sup_count_h"]), w, h, dmin, D, opt.irv_ts,
                                    C.c_float(opt.irv_th), opt.irv_ts if L <= 127 else -1, seed, groups, wpb, 0, stats0) >= 0
        assert same(d0, o["disp_after_irv"])
    a, b = o["disp_after_irv"].copy(), np.empty((h, w), np.float32)
    ms = max(abs(opt.max_disparity), abs(opt.min_disparity))
    emul.emul_interpolate(P(a), P(b), P(o["outlier_label"]), P(left), w, h, 1, ms)
    emul.emul_interpolate(P(b), P(a), P(o["outlier_label"]), P(left), w, h, 2, ms)
    assert same(a, o["disp_after_interp"])
    # the list kernel's walk with empty-space skipping (per-ray step counter, trips of 4 steps, cell-distance skips)
    emul.emul_interpolate_skip.restype = C.c_long
    a2, b2 = o["disp_after_irv"].copy(), np.empty((h, w), np.float32)
    emul.emul_interpolate_skip(P(a2), P(b2), P(o["outlier_label"]), P(left), w, h, 1, ms, 4, None)
    emul.emul_interpolate_skip(P(b2), P(a2), P(o["outlier_label"]), P(left), w, h, 2, ms, 4, None)
    assert same(a2, o["disp_after_interp"])
    # ... and the walk on the padded code map (round 6: k_interpolate_tab as it runs now)
    emul.emul_interpolate_code.restype = C.c_long
    a3, b3 = o["disp_after_irv"].copy(), np.empty((h, w), np.float32)
    assert emul.emul_interpolate_code(P(a3), P(b3), P(o["outlier_label"]), P(left), w, h, 1, ms, 2, 4) >= 0
    assert emul.emul_interpolate_code(P(b3), P(a3), P(o["outlier_label"]), P(left), w, h, 2, ms, 2, 4) >= 0
    assert same(a3, o["disp_after_interp"])
    m = np.empty((h, w), np.float32)
    emul.emul_median_wavefront(P(o["disp_after_dda"]), P(m), w, h)
    assert same(m, o["disp_final"])
    m2 = np.empty((h, w), np.float32)
    if w >= 2 and h >= 2:  # the banded kernel (and its padding rule) is only used for W, H >= 2
        emul.emul_median_padded(P(o["disp_after_dda"]), P(m2), w, h)  # rank selection of the banded kernel
        assert same(m2, o["disp_final"])


def test_gray_all_triples_sample(emul):
    """adc_gray (unfused double arithmetic) == uint8(r*0.299+g*0.587+b*0.114) on a dense sample incl. all greys."""
    rng = np.random.default_rng(7)
    bgr = np.concatenate([rng.integers(0, 256, (400000, 3), dtype=np.uint8),
                          np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1)])
    got = np.empty(len(bgr), np.uint8)
    emul.emul_gray(P(bgr), P(got), C.c_size_t(len(bgr)))
    b, g, r = (bgr[:, i].astype(np.float64) for i in range(3))
    want = ((r * 0.299 + g * 0.587) + b * 0.114).astype(np.uint8)
    assert np.array_equal(got, want)


def test_median9_zero_one_principle(emul):
    """adc_median9 is built from monotone operators (min3/max3/med3): it is the median for all inputs iff it is
    for all 512 0/1 inputs; plus random windows with +-inf and ties."""
    emul.emul_median9.restype = C.c_float
    for bits in range(512):
        v = np.array([(bits >> i) & 1 for i in range(9)], np.float32)
        assert emul.emul_median9(P(v)) == float(np.sort(v)[4]), bits
    rng = np.random.default_rng(3)
    pool = np.array([0.0, 1.0, 1.0, 2.5, np.inf, -np.inf, 7.25, 7.25, 64.0, np.inf], np.float32)
    for _ in range(2000):
        v = rng.choice(pool, 9).astype(np.float32)
        assert emul.emul_median9(P(v)) == float(np.sort(v)[4])


def test_median_padded_matches_reference_small(emul, oracle):
    """wnd[n/2] of the n in-image values == median of nine with (-inf side-centre / +inf corner) padding, on tiny
    images where every pixel is a border pixel, including +inf (invalid) entries."""
    rng = np.random.default_rng(11)
    for (h, w) in [(2, 2), (2, 5), (3, 3), (5, 2), (7, 9)]:
        d = rng.integers(0, 6, (h, w)).astype(np.float32) + rng.integers(0, 2, (h, w)).astype(np.float32) * 0.5
        d[rng.random((h, w)) < 0.25] = np.inf
        want = oracle.median3_inplace(d)
        got = np.empty((h, w), np.float32)
        emul.emul_median_padded(P(d), P(got), w, h)
        assert same(got, want), (h, w)


@pytest.mark.parametrize("kind,w,h,d,seed,rows,depth", [("structured", 200, 300, 32, 5, 64, 1), ("noise", 160, 330, 16, 9, 64, 2),
                                                        ("structured", 240, 330, 32, 6, 64, 2), ("noise", 160, 400, 16, 9, 64, 3),
                                                        ("structured", 200, 300, 32, 5, 8, 1), ("structured", 200, 300, 32, 5, 8, 2)])
def test_median_speculative_bands(emul, port_oracle, kind, w, h, d, seed, rows, depth):
    """k_median_banded with speculative bands (round 4): band b > depth takes its row above from the last link of a chain of copies
    of the bands b - depth .. b - 1, the first of which was filtered starting from the RAW row above it.  With a run-in of 64+
    rows no chain differs from the real band on these maps and the result IS the reference's in-place filter; with 8-row bands
    they do differ -- the count is what the device reports, and the product then redoes the filter in the chained form."""
    from adcensus_amd import workloads
    from oracle import pyoracle
    l, r = workloads.structured_pair(w, h, d, seed=seed) if kind == "structured" else workloads.noise_pair(w, h, seed=seed)
    o = port_oracle.run(l, r, pyoracle.Option(max_disparity=d), stages=["disp_after_interp", "disp_final"])
    raw, out = np.ascontiguousarray(o["disp_after_interp"]), np.empty((h, w), np.float32)
    emul.emul_median_spec_bands.restype = C.c_long
    fails = emul.emul_median_spec_bands(P(raw), P(out), w, h, rows, depth)
    if rows == 64:
        assert fails == 0
    if fails == 0:
        assert same(out, o["disp_final"])
    else:
        assert not same(out, o["disp_final"]) or rows < 64


@pytest.mark.parametrize("kind,w,h,d,seed,rows,depth,nseg,warm", [("structured", 400, 300, 32, 5, 64, 2, 3, 128), ("noise", 336, 330, 16, 9, 64, 2, 2, 128),
                                                                  ("noise", 480, 200, 16, 11, 64, 2, 4, 64), ("structured", 400, 300, 32, 5, 64, 1, 3, 128),
                                                                  ("noise", 336, 330, 16, 9, 64, 2, 3, 0), ("noise", 336, 330, 16, 9, 8, 2, 3, 16),
                                                                  ("structured", 250, 140, 32, 7, 64, 2, 2, 32), ("noise", 130, 70, 16, 3, 64, 2, 2, 16)])
def test_median_speculative_segments(emul, port_oracle, kind, w, h, d, seed, rows, depth, nseg, warm):
    """k_median_banded with speculative COLUMN SEGMENTS (round 6): every band link is cut into segments whose waves run one window of
    levels -- raw values passed through below it (the warm-up starts from unfiltered pixels, like a chain starts from the raw row
    above), nothing computed behind it.  The emulation counts the (band, segment) pairs whose row seam (hand-off consumed over the
    columns xs - 1 .. xe against the map) or column seam (warm-up column xs - 1 against what segment s - 1 wrote) differs -- what
    k_median_seg_check reports.  No failing pair => the map IS the reference's in-place filter (whatever the warm-up was); and no
    value from behind a window's end is ever consumed."""
    from adcensus_amd import workloads
    from oracle import pyoracle
    l, r = workloads.structured_pair(w, h, d, seed=seed) if kind == "structured" else workloads.noise_pair(w, h, seed=seed)
    o = port_oracle.run(l, r, pyoracle.Option(max_disparity=d), stages=["disp_after_interp", "disp_final"])
    raw, out = np.ascontiguousarray(o["disp_after_interp"]), np.empty((h, w), np.float32)
    emul.emul_median_spec_segments.restype = C.c_long
    nan_reads = C.c_long(-1)
    shift = max(0, 2 * rows * depth - warm) & ~15 if (seed % 2) else 0  # (segment 0 narrower, as the launcher does, or equal widths)
    fails = emul.emul_median_spec_segments(P(raw), P(out), w, h, rows, depth, nseg, warm, C.byref(nan_reads), shift)
    assert nan_reads.value == 0
    if fails == 0:
        assert same(out, o["disp_final"])
    if rows == 64 and warm >= 128:
        assert fails == 0
    if warm == 0 and kind == "noise":
        assert fails > 0  # (no warm-up at all: the seams do differ, and the checks see it)


def test_markstein_division_is_ieee_division(tmp_path):
    """The register-ring aggregation divides by the support count with Markstein's sequence on the correctly rounded
    reciprocal (k_aggregate_rr.h: rr_divide).  tools/markstein_check.c compares it with IEEE division over EVERY binary32
    significand; the full sweep over all counts 1..65535 (0 mismatches, ~80 s on 8 cores) is documented in DESIGN.md --
    here: counts 1..96 and the largest ones."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "mk")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-mfma", "-o", exe,
                           os.path.join(root, "tools", "markstein_check.c"), "-lm"])
    for lo, hi in ((1, 96), (4700, 4761), (65500, 65535)):
        out = subprocess.run([exe, str(lo), str(hi)], capture_output=True, text=True, timeout=600).stdout
        assert "mismatches 0" in out, out


@pytest.mark.parametrize("seed,density,ns", [(1, 0.003, 4), (2, 0.02, 4), (3, 0.25, 4), (4, 0.003, 2), (5, 0.0005, 8), (6, 0.0, 4)])
def test_interpolation_skipping_is_exact(emul, seed, density, ns):
    """Empty-space skipping of the ray walk (adc_device_fn.h: adc_itp_*; k_refine.hip): on maps with a nearly empty band (the
    situation the skipping is for: 0.3 % valid pixels), dense regions and empty images, the skipping walk fills every target
    exactly like the plain walk -- and does take fewer look-ups where the map is sparse."""
    rng = np.random.default_rng(seed)
    w, h, ms = 157, 83, 96
    valid = rng.random((h, w)) < density
    valid[:, 100:] |= rng.random((h, w - 100)) < 0.3          # a dense region next to the sparse band
    disp = np.where(valid, rng.integers(0, 90, (h, w)).astype(np.float32) + rng.random((h, w)).astype(np.float32), np.float32(np.inf)).astype(np.float32)
    label = rng.integers(0, 3, (h, w)).astype(np.uint8)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    emul.emul_interpolate_skip.restype = C.c_long
    for which in (1, 2):
        want, got = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
        emul.emul_interpolate(P(disp), P(want), P(label), P(img), w, h, which, ms)
        plain = C.c_long(0)
        n = emul.emul_interpolate_skip(P(disp), P(got), P(label), P(img), w, h, which, ms, ns, C.byref(plain))
        assert same(got, want)
        if density <= 0.003:
            assert n < 0.6 * plain.value, (n, plain.value)   # the sparse band is crossed in jumps


@pytest.mark.parametrize("seed,density,ms,w,h", [(1, 0.003, 96, 157, 83), (2, 0.02, 96, 160, 80), (3, 0.25, 64, 157, 83), (4, 0.003, 2, 8, 8), (5, 0.0005, 200, 131, 97),
                                                 (6, 0.0, 96, 64, 33), (7, 0.22, 128, 203, 61), (8, 0.22, 30, 7, 150), (9, 0.22, 5, 150, 7), (10, 0.1, 1, 97, 31),
                                                 (11, 0.05, 9, 1, 1), (12, 0.3, 600, 50, 40)])
def test_interpolation_on_the_code_map_is_exact(emul, seed, density, ms, w, h):
    """Round 6: the ray walk of k_interpolate_tab reads ONE byte map of the image padded by the search range (adc_device_fn.h:
    valid / outside / skip of the cell) through linear ray offsets, 2 steps in a ray's first trip and 4 in the following ones, without bounds tests.  Every target is filled
    exactly like by the plain walk -- widths that are not multiples of 4, rays that leave the image on every side, ranges from 1 to
    beyond the image size, the validity density of the noise pair -- and every look-up stays inside the padded map."""
    rng = np.random.default_rng(seed)
    valid = rng.random((h, w)) < density
    if w > 100:
        valid[:, 100:] |= rng.random((h, w - 100)) < 0.3
    disp = np.where(valid, rng.integers(0, 90, (h, w)).astype(np.float32) + rng.random((h, w)).astype(np.float32), np.float32(np.inf)).astype(np.float32)
    label = rng.integers(0, 3, (h, w)).astype(np.uint8)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    emul.emul_interpolate_code.restype = C.c_long
    for which in (1, 2):
        want, got = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
        emul.emul_interpolate(P(disp), P(want), P(label), P(img), w, h, which, ms)
        for ns1, ns2 in ((2, 4), (8, 8), (1, 7)):  # (the kernel's trip lengths, the table's limit, an odd pair)
            n = emul.emul_interpolate_code(P(disp), P(got), P(label), P(img), w, h, which, ms, ns1, ns2)
            assert n >= 0, n
            assert same(got, want)


def test_aggregation_launch_gate(emul):
    """agg_gate_skip (k_aggregate_rr.h): of the two plans a mixed stream enqueues back to back (codes 3 / 4, depths packed as
    depth_h | depth_v << 16, 0x7fff = that direction runs the full ring anyway) exactly ONE works for every image; the debug
    surface's per-direction codes 0 / 1 are complementary too; the verify code 2 skips exactly when the arm does not fit."""
    g = emul.emul_agg_gate_skip
    for dh in (1, 3, 8, 0x7fff):
        for dv in (1, 4, 8, 0x7fff):
            thr = dh | (dv << 16)
            for ah in (0, 1, 3, 4, 8, 9, 34):
                for av in (0, 1, 4, 5, 8, 9, 34):
                    for vert in (0, 1):
                        s_skips, f_skips = g(ah, av, 3, thr, vert), g(ah, av, 4, thr, vert)
                        assert s_skips + f_skips == 1, (dh, dv, ah, av)
                        assert s_skips == (0 if (ah <= dh and av <= dv) else 1)
    for small_L in (1, 8):
        for a in (0, 1, 8, 9, 34):
            for vert in (0, 1):
                ah, av = (5, a) if vert else (a, 5)
                assert g(ah, av, 0, small_L, vert) + g(ah, av, 1, small_L, vert) == 1
                assert g(ah, av, 2, small_L, vert) == (1 if a > small_L else 0)
                assert g(ah, av, -1, small_L, vert) == 0


def test_voting_tiles_partition_the_image(emul):
    """irv_plan.h: the tiles of the voting chain's workgroups (bands of IRV_BAND = 8 rows, a band belongs to one XCD) cover every pixel
    exactly once and fit the workgroup's list segment -- for the grids the launcher uses and odd shapes."""
    emul.emul_irv_tile_partition.restype = C.c_long
    for w, h in ((1920, 1080), (1242, 375), (450, 375), (1, 1), (1, 40), (33, 17), (7, 129), (640, 16)):
        for g, wpb in ((128, 8), (256, 8), (1024, 8), (64, 16), (512, 16), (512, 1), (5, 1), (2, 4), (24, 8)):
            for xcd in (0, 1):
                assert emul.emul_irv_tile_partition(w, h, g, wpb, xcd) == 0, (w, h, g, wpb, xcd)


def test_voting_slack_budget_closed_forms(emul):
    """irv_plan.h: irv_level_slack -- the budget of a vote level (how many region pixels may change before the level's outcome can)
    -- is VALID (the level's tests hold at K and below) on a million random levels, and within 3 of the largest valid value."""
    emul.emul_irv_slack_check.restype = C.c_long
    for seed in (1, 2, 3):
        r = emul.emul_irv_slack_check(seed, C.c_long(400000))
        assert r // 1000000 == 0, r  # no invalid budget, ever
        assert r % 1000000 <= 4000, r  # (loose ones: float rounding right at the threshold)


def test_voting_packed_halfword_helpers(emul):
    """irv_plan.h: irv_decode_block (eligible / final / invalid-bin / same-bin masks of 8 packed state halfwords by SWAR
    carries) and the change-tile row test (byte masks from a nibble expansion, any-zero-byte trick) agree with per-pixel
    loops on two million random blocks."""
    emul.emul_irv_swar_check.restype = C.c_long
    for seed in (1, 2):
        assert emul.emul_irv_swar_check(seed, C.c_long(1000000)) == 0


@pytest.mark.parametrize("case", range(10))
def test_voting_chain_slack_budgets_random_cases(emul, port_oracle, case):
    """Round 6: the voting chain with slack budgets (an entry is re-evaluated only when enough pixels of its region's bounding
    rectangle changed to possibly flip its vote, irv_plan.h) on randomly drawn geometries / option sets -- thresholds `irv_ts`
    0..45 and `irv_th` 0.05..0.8 move the budgets through their whole range, arm limits 4..40 the rectangles -- under shuffled
    schedules and three work-list layouts: the result must be the reference's region voting (multistep_refiner.cpp:153-227)."""
    from adcensus_amd import workloads
    from oracle import pyoracle
    rng = np.random.default_rng(6000 + case)
    w, h = int(rng.integers(60, 260)), int(rng.integers(40, 160))
    D = int(rng.choice([16, 32, 64, 100]))
    dmin = int(rng.choice([0, 0, -7, 5]))
    left, right = (workloads.structured_pair(w, h, D, seed=500 + case) if case % 3 else workloads.quantized_noise_pair(w, h, D, seed=500 + case))
    opt = pyoracle.Option(min_disparity=dmin, max_disparity=dmin + D, irv_ts=int(rng.choice([0, 3, 8, 20, 45])),
                          irv_th=float(rng.choice([0.05, 0.2, 0.4, 0.6, 0.8])), cross_L1=int(rng.choice([4, 10, 34, 40])),
                          cross_L2=int(rng.choice([2, 8, 17])), lrcheck_thres=float(rng.choice([0.5, 1.0, 2.0])))
    o = port_oracle.run(left, right, opt)
    L = max(0, min(opt.cross_L1, 255))
    emul.emul_irv_chain2.restype = C.c_long
    evals = {}
    for slack in (8, 1, 40, 0):  # hit entries per wave from which the wave filters (product: 8); 0 = no budgets
        for seed, groups, wpb in ((11, 2, 4), (12, 8, 1), (13, 16, 2)):
            d, stats = o["disp_after_lr"].copy(), (C.c_long * 3)()
            r = emul.emul_irv_chain2(P(d), P(o["outlier_label"]), P(o["arms"]), P(o["sup_count_h"]), w, h, dmin, D, opt.irv_ts,
                                     C.c_float(opt.irv_th), opt.irv_ts if L <= 127 else -1, seed, groups, wpb, slack, stats)
            assert r >= 0, r
            assert same(d, o["disp_after_irv"]), (case, slack, seed)
            evals[(slack, seed)] = stats[1]
    assert sum(v for (s, _), v in evals.items() if s == 1) <= sum(v for (s, _), v in evals.items() if s == 0)
    assert sum(v for (s, _), v in evals.items() if s == 8) <= sum(v for (s, _), v in evals.items() if s == 0)
