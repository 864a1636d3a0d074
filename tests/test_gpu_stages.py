"""GPU tier (T1/T2): every HIP stage against the oracle, stage-isolated, plus the whole Match.
Bit-exact for every buffer (integer maps, f32 volumes, f32 disparity maps)."""
import pytest

from tests import cases, gpu_harness

pytestmark = pytest.mark.gpu

STAGE_CASES = ["s2_96x64_d32", "q_257x131_d64", "q_20x40_d32", "q_9x20_d8", "q_30x7_d8", "q_1x40_d8", "q_40x1_d8",
               "q_3x3_d2", "noise_128x72_d64", "noise_160x90_d128", "noise_150x40_d256", "s2_150x100_neg", "s2_320x180_d128", "s2_200x120_d200", "cone_crop_d40", "cone_crop_L40",
               "cone", "cone_neg", "cone_d16", "cone_nolr", "cone_nofill", "cone_dda", "cone_params",
               # min_disparity > 0, 128 < D < 192 (all-padding chunk), D > 256 up to the maximum of 2047, discontinuity adjustment with dmin != 0
               "cone_pos", "q_40x30_pos_wltd", "noise_160x90_d128_pos", "s2_150x100_pos", "s2_200x120_d160",
               "noise_96x50_d160_neg", "s2_360x60_d300", "noise_80x40_d520", "s2_72x48_d1024", "noise_64x24_d1100", "s2_80x20_d2047",
               "cone_crop_dda_neg", "cone_crop_dda_pos"]


@pytest.mark.parametrize("name", STAGE_CASES)
def test_stage_parity(hip, oracle, name):
    left, right, opt = cases.make_case(name)
    o = oracle.run(left, right, opt)
    rep = gpu_harness.stage_report(left, right, opt, o)
    bad = gpu_harness.failing(rep)
    assert not bad, "%s (oracle=%s): %s" % (name, oracle.kind, bad)


@pytest.mark.parametrize("name", ["cloth3", "piano", "wood2"])
def test_middlebury_pairs(hip, oracle, name):
    """The other pairs of the reference's Data/ directory (committed fixtures tests/golden/<name>_pair.npz); the
    oracle's dumps are additionally checked against the committed SHA-256 table of the reference build."""
    import hashlib
    import json
    import os
    import numpy as np
    left, right, opt = cases.make_case(name)
    o = oracle.run(left, right, opt)
    with open(os.path.join(cases.GOLDEN_DIR, "golden.json")) as f:
        gold = json.load(f)["cases"][name]
    stale = [k for k, v in o.items() if hashlib.sha256(np.ascontiguousarray(cases.canonical(k, v, opt)).tobytes()).hexdigest() != gold[k]]
    assert not stale, "oracle (%s) differs from the committed reference golden: %s" % (oracle.kind, stale)
    rep = gpu_harness.stage_report(left, right, opt, o)
    assert not gpu_harness.failing(rep), gpu_harness.failing(rep)
