"""Failure paths (SURVEY.md 8b "HIP failure -> false"; the reference's contract: bool returns, ADCensusStereo.cpp:31-40,71-76).
libadcensus_hip_faultinj.so is a TEST build of the C ABI (capi.hip with -DADC_FAULT_INJECTION=1 + the product's kernel objects) in
which the n-th HIP call of the object-lifetime / Match path fails; the product library contains no such hook."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAULT_LIB = os.path.join(ROOT, "adcensus_amd", "lib", "libadcensus_hip_faultinj.so")
PROD_LIB = os.path.join(ROOT, "adcensus_amd", "lib", "libadcensus_hip.so")


def _dynamic_symbols(path):
    return subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout


def test_fault_hook_only_in_the_test_build():
    if not os.path.exists(FAULT_LIB):
        pytest.fail("libadcensus_hip_faultinj.so not built (make -C adcensus_amd/csrc)")
    assert "adc_test_fail_at" in _dynamic_symbols(FAULT_LIB)
    prod = _dynamic_symbols(PROD_LIB)
    assert "adc_test_" not in prod
    # ... and the test build exports the whole C ABI (it IS capi.hip)
    import re
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "adcensus_c_api.h")).read(), flags=re.S)
    fault = _dynamic_symbols(FAULT_LIB)
    for name in sorted(set(re.findall(r"\b(adc_[a-z0-9_]+)\s*\(", hdr))):
        assert (" " + name + "\n") in fault, name


@pytest.mark.gpu
def test_hip_failures_surface_as_false_and_leave_the_object_usable(hip):
    env = dict(os.environ, ADC_HIP_LIB=FAULT_LIB, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fault_probe.py")], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("FAULT_PROBE ")][-1]
    o = json.loads(line[len("FAULT_PROBE "):])
    assert o["create_calls"] >= 60 and o["match_calls"] >= 10, o  # (the hook really sits on the paths: allocations, launches, copies)
    assert o["create_not_failed"] == [] and o["create_leaks"] == [] and o["create_past_end_ok"], o
    assert o["match_picks"] >= 10 and o["match_not_failed"] == [] and o["match_wrong_after"] == [], o
    assert o["async_bad"] == [], o
    # farm: submit #2 failed (nonzero), every other pair was delivered with the right map
    assert o["farm_rc"][2] != 0 and [rc for i, rc in enumerate(o["farm_rc"]) if i != 2] == [0] * 5, o
    assert o["farm_delivered"] == 5 and all(ok for i, ok in enumerate(o["farm_outputs_ok"]) if i != 2), o
    assert abs(o["final_leak_bytes"]) <= (2 << 20), o
