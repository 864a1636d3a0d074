"""Named, seeded test cases shared by the golden generator (tools/make_golden.py), the oracle tests
and the GPU parity tests.  A case = (left BGR, right BGR, option)."""
import os

import numpy as np

from adcensus_amd import workloads
from oracle import pyoracle

_HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(_HERE, "golden")


def cone_pair():
    z = np.load(os.path.join(GOLDEN_DIR, "cone_pair.npz"))
    return np.ascontiguousarray(z["left"]), np.ascontiguousarray(z["right"])


def data_pair(name):
    """cloth3 / piano / wood2: the other pairs of the reference's Data/ directory, committed as BGR arrays
    (tests/golden/<name>_pair.npz, written by tools/make_golden.py; PNG decode is lossless)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + "_pair.npz"))
    return np.ascontiguousarray(z["left"]), np.ascontiguousarray(z["right"])


DATA_RANGES = {"cloth3": 128, "piano": 64, "wood2": 128}  # Data/*/d_range.txt


def _crop(pair, y0, y1, x0, x1):
    return tuple(np.ascontiguousarray(a[y0:y1, x0:x1]) for a in pair)


# name -> (builder returning (left,right), option kwargs)
_CASES = {
    # BASELINE.json configs[0]/[1]
    "cone": (cone_pair, dict(max_disparity=64)),
    "cone_neg": (cone_pair, dict(min_disparity=-16, max_disparity=48)),
    "cone_d16": (cone_pair, dict(max_disparity=16)),
    "cone_nolr": (cone_pair, dict(do_lr_check=0)),
    "cone_nofill": (cone_pair, dict(do_filling=0)),
    "cone_dda": (cone_pair, dict(do_discontinuity_adjustment=1)),
    "cone_params": (cone_pair, dict(lambda_ad=7, lambda_census=20, cross_L1=20, cross_L2=9, cross_t1=25, cross_t2=8,
                                    so_p1=0.8, so_p2=2.5, so_tso=11, irv_ts=12, irv_th=0.3, lrcheck_thres=1.5)),
    "cone_crop_d40": (lambda: _crop(cone_pair(), 100, 231, 120, 377), dict(max_disparity=40)),
    # cross_L1 = 40: the aggregation ring (81 entries) does not fit the register ring -> LDS full ring
    "cone_crop_L40": (lambda: _crop(cone_pair(), 100, 231, 120, 377), dict(max_disparity=40, cross_L1=40)),
    # small synthetic cases: odd sizes, W < D, census-skip sizes, 1-pixel-wide/high, VPL = 2 and 4
    "s2_96x64_d32": (lambda: workloads.structured_pair(96, 64, 32, seed=11), dict(max_disparity=32)),
    "s2_320x180_d128": (lambda: workloads.structured_pair(320, 180, 128, seed=12), dict(max_disparity=128)),
    "s2_200x120_d200": (lambda: workloads.structured_pair(200, 120, 200, seed=13), dict(max_disparity=200)),
    "q_257x131_d64": (lambda: workloads.quantized_noise_pair(257, 131, 64, seed=14), dict(max_disparity=64)),
    "q_20x40_d32": (lambda: workloads.quantized_noise_pair(20, 40, 32, seed=15), dict(max_disparity=32)),
    "q_9x20_d8": (lambda: workloads.quantized_noise_pair(9, 20, 8, seed=16), dict(max_disparity=8)),
    "q_30x7_d8": (lambda: workloads.quantized_noise_pair(30, 7, 8, seed=17), dict(max_disparity=8)),
    "q_1x40_d8": (lambda: workloads.quantized_noise_pair(1, 40, 8, seed=18), dict(max_disparity=8)),
    "q_40x1_d8": (lambda: workloads.quantized_noise_pair(40, 1, 8, seed=19), dict(max_disparity=8)),
    "q_3x3_d2": (lambda: workloads.quantized_noise_pair(3, 3, 2, seed=20), dict(max_disparity=2)),
    "noise_128x72_d64": (lambda: workloads.noise_pair(128, 72, seed=21), dict(max_disparity=64)),
    # short arms + D a multiple of 128: small aggregation ring with two disparities per lane, pass pairs (1 and 2 chunks)
    "noise_160x90_d128": (lambda: workloads.noise_pair(160, 90, seed=23), dict(max_disparity=128)),
    "noise_150x40_d256": (lambda: workloads.noise_pair(150, 40, seed=24), dict(max_disparity=256)),
    "s2_150x100_neg": (lambda: workloads.structured_pair(150, 100, 48, seed=22), dict(min_disparity=-8, max_disparity=40)),
    # min_disparity > 0: the right-view WTA, the fused-cost record padding and the scanline
    # interior test all have dmin-dependent branches.  (The reference reads out of bounds in the last min_disparity
    # columns of the right-view map, ADCensusStereo.cpp:296-300: those columns are excluded, see canonical().)
    "cone_pos": (cone_pair, dict(min_disparity=8, max_disparity=72)),
    "q_40x30_pos_wltd": (lambda: workloads.quantized_noise_pair(40, 30, 64, seed=25), dict(min_disparity=5, max_disparity=69)),
    "noise_160x90_d128_pos": (lambda: workloads.noise_pair(160, 90, seed=26), dict(min_disparity=3, max_disparity=131)),
    "s2_150x100_pos": (lambda: workloads.structured_pair(150, 100, 48, seed=27), dict(min_disparity=4, max_disparity=52)),
    # 128 < D < 192: four disparities per lane with a last 64-disparity chunk that is all padding
    "s2_200x120_d160": (lambda: workloads.structured_pair(200, 120, 160, seed=28), dict(max_disparity=160)),
    "noise_96x50_d160_neg": (lambda: workloads.noise_pair(96, 50, seed=29), dict(min_disparity=-70, max_disparity=90)),
    # disparity ranges above 256 (chunked voting histogram, 8 disparities per lane)
    "s2_360x60_d300": (lambda: workloads.structured_pair(360, 60, 300, seed=30), dict(max_disparity=300)),
    "noise_80x40_d520": (lambda: workloads.noise_pair(80, 40, seed=31), dict(min_disparity=-10, max_disparity=510)),
    # the largest range the product accepts (ADC_MAX_DISP_RANGE = 1024: 16 disparities per lane; the voting chain falls back
    # to 8 waves per workgroup because 16 histograms of 1024 bins do not fit into 64 KB of LDS)
    "s2_72x48_d1024": (lambda: workloads.structured_pair(72, 48, 40, seed=32), dict(min_disparity=-512, max_disparity=512)),
    # round 4: ranges above 1024 -- 32 disparities per lane, up to ADC_MAX_DISP_RANGE = 2047 (the 11-bit bins of the voting state map)
    "noise_64x24_d1100": (lambda: workloads.noise_pair(64, 24, seed=33), dict(min_disparity=-40, max_disparity=1060)),
    "s2_80x20_d2047": (lambda: workloads.structured_pair(80, 20, 48, seed=34), dict(min_disparity=-1000, max_disparity=1047)),
    # discontinuity adjustment with min_disparity != 0: the reference indexes the cost row with the ABSOLUTE
    # disparity (multistep_refiner.cpp:331-339), i.e. it reads the neighbouring pixel's costs
    "cone_crop_dda_neg": (lambda: _crop(cone_pair(), 100, 231, 120, 377), dict(min_disparity=-6, max_disparity=40, do_discontinuity_adjustment=1)),
    "cone_crop_dda_pos": (lambda: _crop(cone_pair(), 100, 231, 120, 377), dict(min_disparity=3, max_disparity=43, do_discontinuity_adjustment=1)),
    # the other Middlebury pairs of the reference's Data/ directory, ranges from Data/*/d_range.txt
    "cloth3": (lambda: data_pair("cloth3"), dict(max_disparity=128)),
    "piano": (lambda: data_pair("piano"), dict(max_disparity=64)),
    "wood2": (lambda: data_pair("wood2"), dict(max_disparity=128)),
}
GOLDEN_CASES = list(_CASES.keys())
# subset that the CPU-only tier recomputes with the port (kept small: the whole CPU suite must run in minutes)
FAST_CASES = ["cone_crop_d40", "s2_96x64_d32", "q_257x131_d64", "q_20x40_d32", "q_9x20_d8", "q_30x7_d8", "q_1x40_d8",
              "q_40x1_d8", "q_3x3_d2", "noise_128x72_d64", "s2_150x100_neg", "s2_200x120_d200", "noise_160x90_d128",
              "q_40x30_pos_wltd", "noise_160x90_d128_pos", "s2_150x100_pos", "s2_200x120_d160", "noise_96x50_d160_neg",
              "s2_360x60_d300", "noise_80x40_d520", "s2_72x48_d1024", "noise_64x24_d1100", "s2_80x20_d2047", "cone_crop_dda_neg", "cone_crop_dda_pos"]


def canonical(stage, arr, opt):
    """The part of a stage dump that is defined behaviour of the reference: for min_disparity > 0 the right-view
    WTA map's last min_disparity columns come from an out-of-bounds read (ADCensusStereo.cpp:296-300 with
    best_disparity still 0) and are left out of hashes and comparisons.  (They are never consumed: the LR check
    reads column lround(x - d) <= W - 1 - min_disparity.)"""
    if stage == "disp_right_wta" and opt.min_disparity > 0:
        return np.ascontiguousarray(arr[:, :max(0, arr.shape[1] - opt.min_disparity)])
    return arr


def make_case(name):
    build, kw = _CASES[name]
    left, right = build()
    return left, right, pyoracle.Option(**kw)


def to_product_option(opt):
    """pyoracle.Option -> adcensus_amd.ADCensusOption (identical layout)."""
    import ctypes as C
    from adcensus_amd import ADCensusOption
    o = ADCensusOption()
    C.memmove(C.byref(o), C.byref(opt), C.sizeof(o))
    return o
