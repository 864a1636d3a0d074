"""The register-ring aggregation kernels (k_agg_regring*) keep their ring in VGPRs v56..v127 (pass pairs: a second ring in
v128..v199) that only inline asm touches.  That is sound only while the COMPILER never allocates one of those registers in these kernels: this test
compiles k_aggregate.hip to assembly with the product flags and checks every instruction outside the asm blocks."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
RING_V0 = 56


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_compiler_stays_below_the_ring_registers(tmp_path):
    src = os.path.join(ROOT, "adcensus_amd", "csrc", "k_aggregate.hip")
    out = str(tmp_path / "k_aggregate.s")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
           "-Wno-inline-asm", "-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    text = open(out).read()
    names = re.findall(r"^(_Z\d+k_agg_regring\w*):", text, re.M)
    assert len(names) >= 7, names  # 4 plain passes + the fused-cost pass + 2 pass pairs
    for name in names:
        a = re.search(r"^" + re.escape(name) + r":", text, re.M).start()
        body = text[a:text.index("s_endpgm", a)]
        in_asm, worst = False, -1
        for line in body.split("\n"):
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            if in_asm:
                continue
            code = line.split(";")[0]
            regs = [int(r) for r in re.findall(r"\bv(\d+)\b", code)] + [int(hi) for _, hi in re.findall(r"\bv\[(\d+):(\d+)\]", code)]
            worst = max([worst] + regs)
        assert 0 <= worst < RING_V0, "%s: the compiler uses v%d (ring starts at v%d)" % (name, worst, RING_V0)
        m = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
        want = 200 if "regring_pair" in name else 128
        assert m and re.search(r"\.amdhsa_next_free_vgpr %d\b" % want, m.group(1)), "%s: kernel descriptor must reserve %d VGPRs" % (name, want)
    shutil.rmtree(str(tmp_path), ignore_errors=True)
