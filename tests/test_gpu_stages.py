"""GPU tier (T1/T2): every HIP stage against the oracle, stage-isolated, plus the whole Match.
Bit-exact for every buffer (integer maps, f32 volumes, f32 disparity maps)."""
import pytest

from tests import cases, gpu_harness

pytestmark = pytest.mark.gpu

STAGE_CASES = ["s2_96x64_d32", "q_257x131_d64", "q_20x40_d32", "q_9x20_d8", "q_30x7_d8", "q_1x40_d8", "q_40x1_d8",
               "q_3x3_d2", "noise_128x72_d64", "noise_160x90_d128", "noise_150x40_d256", "s2_150x100_neg", "s2_320x180_d128", "s2_200x120_d200", "cone_crop_d40", "cone_crop_L40",
               "cone", "cone_neg", "cone_d16", "cone_nolr", "cone_nofill", "cone_dda", "cone_params"]


@pytest.mark.parametrize("name", STAGE_CASES)
def test_stage_parity(hip, oracle, name):
    left, right, opt = cases.make_case(name)
    o = oracle.run(left, right, opt)
    rep = gpu_harness.stage_report(left, right, opt, o)
    bad = gpu_harness.failing(rep)
    assert not bad, "%s (oracle=%s): %s" % (name, oracle.kind, bad)


@pytest.mark.parametrize("name", ["cloth3", "piano", "wood2"])
def test_middlebury_pairs(hip, oracle, name):
    """The other pairs of the reference's Data/ directory (fixtures travel in tests/golden/_data)."""
    pair = cases.data_pair(name)
    if pair is None:
        pytest.skip("tests/golden/_data/%s_pair.npz not present" % name)
    from oracle import pyoracle
    opt = pyoracle.Option(max_disparity=cases.DATA_RANGES[name])
    o = oracle.run(pair[0], pair[1], opt)
    rep = gpu_harness.stage_report(pair[0], pair[1], opt, o)
    assert not gpu_harness.failing(rep), gpu_harness.failing(rep)
