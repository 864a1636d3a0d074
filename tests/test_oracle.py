"""CPU tier: pins the oracles.

* the golden SHA-256 table tests/golden/golden.json was produced by the REAL reference build
  (oracle/_ref) in the build container (tools/make_golden.py);
* the plain-C port must reproduce every stage dump of every golden case bit-for-bit;
* when oracle/_ref is present it is re-checked against the same table (recipe drift guard).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import cases

with open(os.path.join(cases.GOLDEN_DIR, "golden.json")) as f:
    GOLDEN = json.load(f)["cases"]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _check(oracle, name):
    left, right, opt = cases.make_case(name)
    dumps = oracle.run(left, right, opt)
    bad = [k for k, v in dumps.items() if _sha(cases.canonical(k, v, opt)) != GOLDEN[name][k]]
    assert not bad, "%s oracle differs from the reference golden on %s: %s" % (oracle.kind, name, bad)


def test_golden_table_complete():
    assert set(GOLDEN.keys()) == set(cases.GOLDEN_CASES)


@pytest.mark.parametrize("name", cases.FAST_CASES + ["cone"])
def test_port_matches_reference_golden(port_oracle, name):
    _check(port_oracle, name)


@pytest.mark.parametrize("name", ["cone", "s2_96x64_d32", "q_20x40_d32"])
def test_ref_build_matches_golden(ref_oracle, name):
    if ref_oracle is None:
        pytest.skip("oracle/_ref not built and /root/reference absent")
    _check(ref_oracle, name)


def test_oracle_initialize_contract(port_oracle, ref_oracle):
    """Initialize -> false for w,h <= 0 or empty disparity range (ADCensusStereo.cpp:31-40)."""
    from oracle import pyoracle
    img = np.zeros((4, 4, 3), np.uint8)
    for orc in [o for o in (port_oracle, ref_oracle) if o is not None]:
        with pytest.raises(RuntimeError):
            orc.run(img, img, pyoracle.Option(min_disparity=5, max_disparity=5))
        with pytest.raises(RuntimeError):
            orc.run(img, img, pyoracle.Option(min_disparity=9, max_disparity=3))


def test_median_is_recursive(port_oracle):
    """The 3x3 median runs in place (adcensus_util.cpp:55-81 with in==out): it must differ from an
    out-of-place median on a generic map -- guards against 'fixing' the oracle."""
    rng = np.random.default_rng(0)
    d = rng.uniform(0, 60, (40, 50)).astype(np.float32)
    rec = port_oracle.median3_inplace(d)
    pad = np.pad(d, 1, constant_values=np.nan)
    win = np.stack([pad[r:r + 40, c:c + 50] for r in range(3) for c in range(3)], -1)
    interior = np.median(win[1:-1, 1:-1], axis=-1)
    assert (rec[1:-1, 1:-1] != interior).mean() > 0.1
