"""GPU tier: the OPT-IN paper modes (adc_set_paper_modes) -- 5x5 census, averaged scanline paths,
right-image arms.  They are not the reference's behaviour: the checker is the port oracle's own restatement of the same
definitions (oracle/adcensus_port.c: adc_oracle_run_paper), compared stage by stage and through the whole Match, bit for
bit; and with the modes off the same handle gives the reference result again."""
import numpy as np
import pytest

from tests import cases, gpu_harness

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["s2_96x64_d32", "q_20x40_d32", "cone_crop_d40", "s2_320x180_d128", "s2_150x100_neg", "q_9x20_d8"])
@pytest.mark.parametrize("modes", [1, 2, 4, 7])
def test_paper_modes_match_the_port_oracle(hip, port_oracle, name, modes):
    left, right, opt = cases.make_case(name)
    o = port_oracle.run(left, right, opt, paper_modes=modes)
    rep = gpu_harness.stage_report(left, right, opt, o, paper_modes=modes)
    bad = gpu_harness.failing(rep)
    assert not bad, "%s modes=%d: %s" % (name, modes, bad)


def test_paper_modes_change_results_and_switch_off(hip, port_oracle):
    A = hip
    left, right, opt = cases.make_case("s2_96x64_d32")
    base = port_oracle.run(left, right, opt, stages=["disp_final"])["disp_final"]
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(left.shape[1], left.shape[0], cases.to_product_option(opt))
    assert np.array_equal(st.match(left, right).view(np.uint32), base.view(np.uint32))
    for modes in (A.PAPER_CENSUS5X5, A.PAPER_SO_SUM, A.PAPER_RIGHT_ARMS):
        st.set_paper_modes(modes)
        want = port_oracle.run(left, right, opt, stages=["disp_final"], paper_modes=modes)["disp_final"]
        got = st.match(left, right)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert not np.array_equal(got.view(np.uint32), base.view(np.uint32))  # a different algorithm by definition
    st.set_paper_modes(0)
    assert np.array_equal(st.match(left, right).view(np.uint32), base.view(np.uint32))  # off again: the reference's result
    with pytest.raises(RuntimeError):
        st.set_paper_modes(64)
    st.Release()
