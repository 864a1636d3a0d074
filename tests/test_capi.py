"""CPU tier: the C-ABI library loads and exports every symbol include/adcensus_c_api.h declares; the
host-side contract that needs no GPU (argument validation = the reference's `false` returns)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "adcensus_c_api.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(adc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    import adcensus_amd
    lib = adcensus_amd.lib()
    names = _declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_option_defaults_match_reference():
    """adcensus_types.h:67-74"""
    from adcensus_amd import ADCensusOption
    o = ADCensusOption()
    assert (o.min_disparity, o.max_disparity, o.lambda_ad, o.lambda_census) == (0, 64, 10, 30)
    assert (o.cross_L1, o.cross_L2, o.cross_t1, o.cross_t2) == (34, 17, 20, 6)
    assert (o.so_p1, o.so_p2, o.so_tso, o.irv_ts) == (1.0, 3.0, 15, 20)
    assert abs(o.irv_th - 0.4) < 1e-7 and o.lrcheck_thres == 1.0
    assert (o.do_lr_check, o.do_filling, o.do_discontinuity_adjustment) == (1, 1, 0)
    assert C.sizeof(o) == 60


def test_initialize_rejects_bad_arguments():
    """Initialize -> false on w,h<=0 or empty range (ADCensusStereo.cpp:31-40); checked before any HIP call."""
    import adcensus_amd
    st = adcensus_amd.ADCensusStereo()
    assert not st.Initialize(0, 10, adcensus_amd.ADCensusOption())
    assert not st.Initialize(10, -1, adcensus_amd.ADCensusOption())
    assert not st.Initialize(10, 10, adcensus_amd.ADCensusOption(min_disparity=4, max_disparity=4))
    assert not st.Initialize(10, 10, adcensus_amd.ADCensusOption(min_disparity=0, max_disparity=adcensus_amd.MAX_DISP_RANGE + 1))
    # Match before a successful Initialize -> false (ADCensusStereo.cpp:71-73)
    import numpy as np
    assert not st.Match(np.zeros((10, 10, 3), np.uint8), np.zeros((10, 10, 3), np.uint8), np.zeros((10, 10), np.float32))


def test_no_cpu_fallback_without_device():
    """Without a GPU the product must FAIL (never silently compute on the CPU)."""
    import adcensus_amd
    if adcensus_amd.device_count() > 0:
        pytest.skip("a GPU is visible here")
    st = adcensus_amd.ADCensusStereo()
    assert not st.Initialize(32, 32, adcensus_amd.ADCensusOption())


def test_product_does_not_reference_oracle():
    """The product sources must not import / link / call anything under oracle/."""
    bad = []
    for base in ("adcensus_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".c", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|oracle/|pyoracle|libadcensus_(ref|port)", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_types_header_is_source_compatible(tmp_path):
    """include/adcensus_types.h provides every name the reference's adcensus_types.h puts into the global namespace
    (adcensus_types.h:12-17,21-42,45-86): a translation unit that uses all of them compiles against include/ alone."""
    import subprocess
    src = tmp_path / "names.cpp"
    src.write_text(
        '#include "ADCensusStereo.h"\n'
        "static_assert(sizeof(sint8) == 1 && sizeof(uint16) == 2 && sizeof(sint64) == 8 && sizeof(float64) == 8, \"\");\n"
        "int main() {\n"
        "  ADColor c(1, 2, 3); CensusSize s = Census9x7; float32* p = new float32[3]; SAFE_DELETE(p);\n"
        "  vector<pair<sint32, sint32>> v; ADCensusOption o; ADCensusStereo st;\n"
        "  const bool ok = c.b == 1 && c.g == 2 && c.r == 3 && s == 1 && Census5x5 == 0 && !p && v.empty() && o.cross_L1 == 34 &&\n"
        "                  Large_Float == 99999.0f && Small_Float == -99999.0f && Invalid_Float > Large_Float;\n"
        "  (void)st; return ok ? 0 : 1;\n"
        "}\n")
    subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])


def _cli():
    cli = os.path.join(ROOT, "adcensus_amd", "bin", "adcensus_cli")
    if not os.path.exists(cli):
        pytest.fail("adcensus_cli not built (python -c 'import __graft_entry__ as g; g.build()')")
    return cli


def test_cli_png_codec_and_jet_colormap(tmp_path):
    """The CLI's own PNG reader / writer (main.cpp:47-48,203-209 use OpenCV) round-trips the reference's image types, and its
    COLORMAP_JET table reproduces the reference's colour-mapped result image from its grey one, pixel for pixel."""
    import subprocess
    import numpy as np
    from PIL import Image
    from tests import cases
    cli = _cli()
    left, _ = cases.cone_pair()
    rgb = np.ascontiguousarray(left[:, :, ::-1])
    for name, im in (("rgb.png", Image.fromarray(rgb)), ("gray.png", Image.fromarray(rgb[:, :, 1])),
                     ("pal.png", Image.fromarray(rgb).quantize(64)), ("rgba.png", Image.fromarray(np.dstack([rgb, rgb[:, :, :1]])))):
        src, dst = str(tmp_path / name), str(tmp_path / ("out_" + name))
        im.save(src)
        subprocess.check_call([cli, "--convert", src, dst])
        want = np.array(Image.open(src).convert("RGB"))
        got = np.array(Image.open(dst))
        assert got.shape == want.shape and np.array_equal(got, want), name
    d = os.path.join(ROOT, "tests", "golden", "ref_cone-d.png")  # the reference's own result images (doc/exp/res)
    c = os.path.join(ROOT, "tests", "golden", "ref_cone-c.png")
    out = str(tmp_path / "c.png")
    subprocess.check_call([cli, "--colormap", d, out])
    assert np.array_equal(np.array(Image.open(out)), np.array(Image.open(c).convert("RGB")))
