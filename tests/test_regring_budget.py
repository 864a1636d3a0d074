"""The register-ring aggregation kernels keep their ring in VGPRs that only inline asm touches: k_agg_regring* v56..v127
(pass pairs: a second ring in v128..v199), k_agg_rr2* v96..v239 (ring slots = VGPR pairs).  That is sound only while the
COMPILER never allocates one of those registers in these kernels, never spills, and emits exactly the vector-memory
operations per step the hand-counted s_waitcnt vmcnt values assume: this test compiles k_aggregate.hip to assembly with
the product flags and checks every instruction outside the asm blocks, the kernel descriptors and the steady-state loop."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
RING_V0 = 56


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_compiler_stays_below_the_ring_registers(device_asm):
    text = open(device_asm("k_aggregate")).read()
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
    # ---- third generation: ring of VGPR pairs v96..v239
    names2 = re.findall(r"^(_Z\d+k_agg_rr2\w*):", text, re.M)
    assert len(names2) >= 5, names2  # 4 plain passes + the fused-cost pass
    for name in names2:
        a = re.search(r"^" + re.escape(name) + r":", text, re.M).start()
        body = text[a:text.index("s_endpgm", a)]
        in_asm, worst = False, -1
        steady = []  # compiler-issued vector-memory operations between two consecutive steady-state waits
        cur = None
        for line in body.split("\n"):
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            code = line.split(";")[0]
            if in_asm:
                if "s_waitcnt vmcnt(14)" in code:
                    if cur is not None:
                        steady.append(cur)
                    cur = 0
                continue
            regs = [int(r) for r in re.findall(r"\bv(\d+)\b", code)] + [int(hi) for _, hi in re.findall(r"\bv\[(\d+):(\d+)\]", code)]
            worst = max([worst] + regs)
            assert "scratch_" not in code and "buffer_store" not in code and "buffer_load" not in code, "%s: spill / scratch access: %s" % (name, code)
            if cur is not None and re.search(r"\bglobal_(load|store)", code):
                cur += 1
            # an unconditional branch ends the fall-through path: what follows textually (a rotated loop's latch block in front of
            # its header) is reached from elsewhere, with its own count -- the path-exact proof is tools/check_async_loads.py
            if cur is not None and re.match(r"\s*s_branch\b", code):
                cur = 0
        assert 0 <= worst < 96, "%s: the compiler uses v%d (ring starts at v96)" % (name, worst)
        m = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
        assert m and re.search(r"\.amdhsa_next_free_vgpr 240\b", m.group(1)), "%s: kernel descriptor must reserve 240 VGPRs (2 waves per SIMD)" % name
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", m.group(1)) or "private_segment_fixed_size" not in m.group(1), name
        if "cost" not in name:
            # the steady-state wait vmcnt(14) assumes EXACTLY one compiler-issued vector-memory operation (the output store)
            # per step next to the asm-issued prefetch load; (9 such waits: the last step of the peeled block + the 8 of the loop body; the last interval runs into the drain code)
            assert len(steady) >= 8 and all(c == 1 for c in steady[:7]), "%s: vector-memory operations per steady-state step: %s" % (name, steady[:20])
