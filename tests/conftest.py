import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(target_dir, *args):
    subprocess.check_call(["make", "-C", target_dir, *args], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def port_oracle():
    """oracle/adcensus_port.c (plain-C restatement); built on demand."""
    from oracle import pyoracle
    if not pyoracle.have_port():
        _make(os.path.join(ROOT, "oracle"), "port")
    return pyoracle.load("port")


@pytest.fixture(scope="session")
def ref_oracle():
    """oracle/_ref (the reference's own sources); None when neither built nor buildable."""
    from oracle import pyoracle
    if not pyoracle.have_ref() and os.path.isdir("/root/reference/AD-Census"):
        _make(os.path.join(ROOT, "oracle"), "ref")
    return pyoracle.load("reference") if pyoracle.have_ref() else None


@pytest.fixture(scope="session")
def oracle(ref_oracle, port_oracle):
    """The checker used by the GPU parity tests: the real reference build when present, else the port
    (which the CPU tier pins against the reference's golden hashes)."""
    return ref_oracle or port_oracle


@pytest.fixture(scope="session")
def emul():
    """tests/emul/emul.cpp -> scalar CPU emulation of the kernels' parallel formulations."""
    import ctypes
    out_dir = os.path.join(ROOT, "tests", "emul", "_build")
    so = os.path.join(out_dir, "libadcensus_emul.so")
    srcs = [os.path.join(ROOT, "tests", "emul", "emul.cpp"), os.path.join(ROOT, "tests", "emul", "emul_rr.cpp"),
            os.path.join(ROOT, "tests", "emul", "emul_irv.cpp")]
    hdrs = [os.path.join(ROOT, "adcensus_amd", "csrc", "adc_device_fn.h"), os.path.join(ROOT, "adcensus_amd", "csrc", "k_aggregate_rr.h"),
            os.path.join(ROOT, "adcensus_amd", "csrc", "k_aggregate_rr2.h"),
            os.path.join(ROOT, "adcensus_amd", "csrc", "irv_plan.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs + hdrs):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wno-unknown-pragmas", "-fPIC", "-shared", *srcs, "-o", so])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests fail loudly (not skip) when it is missing or no device is visible."""
    import adcensus_amd
    adcensus_amd.lib()
    if adcensus_amd.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests must run on a MI355X box")
    return adcensus_amd


HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="session")
def device_asm(tmp_path_factory):
    """device_asm("k_scanline") -> path of the gfx950 assembly of adcensus_amd/csrc/k_scanline.hip, compiled once per session
    with the product flags (csrc/Makefile) -- what the generated-code checks (register budgets, asynchronous loads) read."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out_dir = tmp_path_factory.mktemp("device_asm")
    cache = {}

    def get(stem):
        if stem not in cache:
            src = os.path.join(ROOT, "adcensus_amd", "csrc", stem + ".hip")
            out = str(out_dir / (stem + ".s"))
            cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                   "-Wno-inline-asm", "-Wno-unused-value", "-S", "--cuda-device-only", src, "-o", out]
            subprocess.run(cmd, check=True, capture_output=True, timeout=900)
            cache[stem] = out
        return cache[stem]

    return get
