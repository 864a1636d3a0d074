"""GPU tier: the last aggregation pass inside the L->R scanline pass (k_scanline_seg_agg, round 5).  The form runs from the SECOND
Match of a handle on (the arm depth is assumed from the previous image), for short-arm images with two disparities per lane whose
row passes run as verified segments of the compiler-allocated kernel family -- so every case here matches twice or more on one
handle, checks the result against the CPU oracle bit for bit and asserts that the fused form really ran (debug counter 13)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from adcensus_amd import workloads
from oracle import pyoracle
from tests import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


@pytest.mark.parametrize("w,h,dmin,d", [(640, 520, 0, 128), (1242, 375, 0, 128), (400, 1100, -10, 128), (901, 515, 5, 100)])
def test_fused_tail_geometries(hip, oracle, w, h, dmin, d):
    """Short-arm (noise) pairs at sizes where the row passes run as segments of k_scanline_seg: odd width, a tall image, negative and
    positive min_disparity, a range with padding lanes (100 of 128).  Two different pairs alternate on one handle."""
    A = hip
    opt = pyoracle.Option(min_disparity=dmin, max_disparity=dmin + d)
    pairs = [workloads.noise_pair(w, h, seed=9100 + k) for k in range(2)]
    want = [oracle.run(l, r, opt, stages=["disp_final"])["disp_final"] for l, r in pairs]
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(w, h, cases.to_product_option(opt))
    for k in (0, 0, 1, 0, 1):
        got = st.match(*pairs[k])
        assert _same(got, want[k]), "%dx%d [%d, %d): Match of pair %d differs in %d pixels" % (
            w, h, dmin, dmin + d, k, int((got.view(np.uint32) != want[k].view(np.uint32)).sum()))
    assert st.debug_counter(13) >= 3, "the fused form did not run (%d)" % st.debug_counter(13)
    assert st.debug_counter(2) == 0 and st.debug_counter(4) == 0  # no redo
    st.Release()


def test_fused_tail_switched_off_is_identical(hip):
    """ADC_FUSE_AGG_SO=0 (own interpreter: the switch is read once) gives the same maps as the fused form, pair by pair."""
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "import adcensus_amd as A\n"
            "from adcensus_amd import workloads\n"
            "st = A.ADCensusStereo(device=0); assert st.Initialize(800, 600, A.ADCensusOption(max_disparity=128))\n"
            "out = []\n"
            "for k in (0, 1, 2, 1):\n"
            "    out.append(hashlib.sha256(st.match(*workloads.noise_pair(800, 600, seed=777 + k)).tobytes()).hexdigest()[:16])\n"
            "print('DIGESTS', ' '.join(out), 'FUSED', st.debug_counter(13))\n") % ROOT
    res = {}
    for flag in ("1", "0"):
        o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ADC_FUSE_AGG_SO=flag), capture_output=True, text=True, timeout=600)
        assert o.returncode == 0, o.stdout[-1500:] + o.stderr[-1500:]
        line = [l for l in o.stdout.splitlines() if l.startswith("DIGESTS")][-1].split()
        res[flag] = (line[1:5], int(line[-1]))
    assert res["1"][0] == res["0"][0], res
    assert res["1"][1] >= 3 and res["0"][1] == 0, res


@pytest.mark.parametrize("env,expect_redo", [({"ADC_SO_FAST": "0"}, False), ({"ADC_SO_FAST": "0", "ADC_SO_SEG": "3", "ADC_SO_WARM": "16"}, True)])
def test_fused_tail_small_images_and_failing_seams(hip, env, expect_redo):
    """With the compiler-allocated kernel family forced (ADC_SO_FAST=0) small images run the fused form too: the random geometries /
    options of tests/test_gpu_random.py once more in that setting; and with a 16-step warm-up seams DO fail -- adc_wait then redoes
    from the aggregation on with the full ring and whole rows (the redo never takes the fused form), and the handle keeps whole rows
    afterwards (so it does not fuse either): same maps."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "import adcensus_amd as A\n"
            "from oracle import pyoracle\n"
            "from tests import cases, test_gpu_random\n"
            "orc = pyoracle.load('auto')\n"
            "bad, fused, redos = [], 0, 0\n"
            "rng = np.random.default_rng(4711)\n"
            "for k in range(10):\n"
            "    (l, r), opt = test_gpu_random._draw(rng)\n"
            "    if k %% 2: l, r = (a.copy() for a in __import__('adcensus_amd').workloads.noise_pair(l.shape[1], l.shape[0], seed=50 + k))\n"
            "    want = orc.run(l, r, opt, stages=['disp_final'])['disp_final']\n"
            "    st = A.ADCensusStereo(device=0)\n"
            "    assert st.Initialize(l.shape[1], l.shape[0], cases.to_product_option(opt))\n"
            "    for rep in range(3):\n"
            "        d = st.match(l, r)\n"
            "        if not np.array_equal(d.view(np.uint32), want.view(np.uint32)): bad.append((k, rep, l.shape, opt.min_disparity, opt.max_disparity))\n"
            "    fused += st.debug_counter(13); redos += st.debug_counter(4)\n"
            "    st.Release()\n"
            "print('BAD', bad, 'FUSED', fused, 'SEAM_REDOS', redos)\n"
            "sys.exit(1 if bad else (3 if %r and redos == 0 else (2 if not %r and fused == 0 else 0)))\n") % (ROOT, expect_redo, expect_redo)
    o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert o.returncode == 0, o.stdout[-2000:] + o.stderr[-2000:]
