"""K4, K5 and K11 issue their prefetch loads from inline asm into registers the COMPILER allocates and wait for them with
hand-counted `s_waitcnt vmcnt(N)`.  That is sound only while (i) every hand-counted wait really covers the load it takes over
and (ii) the compiler never copies, spills or re-uses such a register while its load is in flight (it cannot know: for the
compiler the value exists as soon as the asm statement has been issued).  tools/check_async_loads.py proves both on the
generated code: a lower bound of the number of younger vector-memory operations per in-flight register, propagated over the
control-flow graph (branches on wave-uniform constants followed exactly).  Every kernel must be clean.  (Found with it: a scanline variant with two steady-state forms, for which the register allocator rotated the
prefetch slots and copied in-flight registers at the loop back edge -- tools/experiments/scanline_interior_chunks.patch.)"""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("check_async_loads", os.path.join(ROOT, "tools", "check_async_loads.py"))
cal = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(cal)

# (Kernels whose findings would be infeasible paths of the analysis -- none at present: branches on wave-uniform constants, the
# compiler's encoding of "exit path shares a block with the back edge" and of if / else with a common tail, are followed exactly.)
KNOWN_IMPRECISE = ()


def _check(path, want):
    res = cal.check_file(path)
    seen = {w: 0 for w in want}
    problems = []
    for name, r in res.items():
        for w in want:
            if w in name:
                seen[w] += 1
        if any(k in name for k in KNOWN_IMPRECISE):
            continue
        if r["bad"]:
            problems.append("%s: %d violations, first: %s" % (name, len(r["bad"]), r["bad"][0]))
    assert not problems, "\n".join(problems)
    return res, seen


def test_scanline_prefetch_slots(device_asm):
    res, seen = _check(device_asm("k_scanline"), ["k_scanlineILi1E", "k_scanlineILi2E", "k_scanline_pinILi1E", "k_scanline_pinILi2E",
                                                  "k_scanline_segILi1E", "k_scanline_segILi2E", "k_scanline_pin_segILi1E", "k_scanline_pin_segILi2E",
                                                  "k_scanline_seg_aggILi2E"])  # (+ the L->R pass that also aggregates its input)
    # every asm-prefetch instantiation of both kernel families was analysed (+ the row passes cut into verified segments)
    assert all(v == (1 if "_seg" in k else 5) for k, v in seen.items()), seen
    assert all(r["asm_loads"] >= 100 for r in res.values())  # prologue + first block + both steady-state forms, 16 slots + d1 words
    # k_scanline_pin: the slots are registers the compiler cannot allocate (amdgpu_num_vgpr(96) + named registers v96..v147): every
    # instruction outside the asm statements stays below v96, nothing is spilled, and the descriptor reserves 148 registers
    text = open(device_asm("k_scanline")).read()
    for name, body in cal.functions(text):
        if not re.search(r"k_scanline_pin(_seg)?ILi[12]E", name):
            continue
        in_asm, worst = False, -1
        for line in body:
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if in_asm or not t or t[0] in ";.":
                continue
            code = t.split(";")[0]
            assert "scratch_" not in code, "%s: spill: %s" % (name, code)
            worst = max([worst] + list(cal.regs(code)))
        assert 0 <= worst < 96, "%s: the compiler uses v%d (slots start at v96)" % (name, worst)
        m = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
        assert m and re.search(r"\.amdhsa_next_free_vgpr 148\b", m.group(1)), name


def test_median_window_prefetch(device_asm):
    res, seen = _check(device_asm("k_refine"), ["k_median_bandedILb1E", "k_median_bandedILb0E"])
    assert seen["k_median_bandedILb1E"] == 1 and seen["k_median_bandedILb0E"] == 1, seen


def test_right_wta_march_prefetch(device_asm):
    """K6 marching kernel: the pixel vectors of the next two steps are in flight; ONE hand-counted wait per step (vmcnt = the
    loads of one step) takes a step over.  A wait one load too weak must be reported."""
    res, seen = _check(device_asm("k_wta"), ["k_wta_right_marchILi1E", "k_wta_right_marchILi2E"])
    assert seen == {"k_wta_right_marchILi1E": 1, "k_wta_right_marchILi2E": 1}, seen
    assert sorted(r["asm_loads"] for r in res.values()) == [32, 64]  # two steps before the loop + two in it, 8 * VPL loads each
    text = open(device_asm("k_wta")).read()
    for pat, old, new in (("k_wta_right_marchILi1E", "vmcnt(8)", "vmcnt(9)"), ("k_wta_right_marchILi2E", "vmcnt(16)", "vmcnt(17)")):
        for name, body in cal.functions(text):
            if pat in name:
                weak = [ln.replace(old, new) for ln in body]
                assert weak != body and cal.analyse(weak)["bad"], (name, old)
        # no scratch, and few enough registers for a workgroup of 8 waves
        m = re.search(r"\.amdhsa_kernel _Z\d+" + pat + r"\w*\n(.*?)\.end_amdhsa_kernel", text, re.S)
        assert m and re.search(r"\.amdhsa_private_segment_fixed_size 0\b", m.group(1)), pat
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(1)).group(1)) <= 128, pat


def test_aggregation_prefetch_and_record_blocks(device_asm):
    res, seen = _check(device_asm("k_aggregate"), ["k_agg_march", "k_agg_rr2I", "k_agg_rr2_cost", "k_agg_regringI", "k_agg_regring_cost",
                                                   "k_agg_regring_pair"])
    assert seen["k_agg_march"] >= 20 and seen["k_agg_rr2I"] == 4 and seen["k_agg_rr2_cost"] == 1, seen
    assert seen["k_agg_regringI"] == 4 and seen["k_agg_regring_cost"] == 1 and seen["k_agg_regring_pair"] == 2, seen
    # the 64-entry record blocks of the fused-cost pass are taken over without a wait (>= 64 younger operations by construction)
    assert [r for n, r in res.items() if "k_agg_rr2_cost" in n][0]["deferred"] >= 1


def test_checker_catches_weakened_waits_in_the_aggregation_kernels(device_asm):
    text = open(device_asm("k_aggregate")).read()
    for pat, old, new in (("k_agg_regring_pairILb1E", "vmcnt(8)", "vmcnt(9)"), ("k_agg_rr2ILb1ELb1E", "vmcnt(14)", "vmcnt(16)"),
                          ("k_agg_marchILb1ELb1ELb1ELb0ELb1ELi2E", "vmcnt(14)", "vmcnt(16)")):
        for name, body in cal.functions(text):
            if pat in name:
                assert not cal.analyse(body, allow_deferred=True)["bad"], name
                weak = [ln.replace(old, new) for ln in body]
                assert weak != body and cal.analyse(weak, allow_deferred=True)["bad"], (name, old)


def test_checker_catches_a_weakened_wait_and_a_slot_copy(device_asm, tmp_path):
    text = open(device_asm("k_scanline")).read()
    weak = str(tmp_path / "weak.s")
    open(weak, "w").write(text.replace("s_waitcnt vmcnt(50)", "s_waitcnt vmcnt(52)"))
    res = cal.check_file(weak, "k_scanline(_pin)?ILi2ELb0ELb1ELb0E")
    assert len(res) == 2 and all(r["bad"] for r in res.values()), "a wait two operations too weak must be reported"
    # a compiler-style copy of a slot register right after its load has been issued
    m = re.search(r"(\tglobal_load_dwordx2 (v\[\d+:\d+\]), v\[\d+:\d+\], off(?: nt)?\n\t;;#ASMEND\n)", text)  # a compiler-allocated slot
    assert m
    copy = str(tmp_path / "copy.s")
    open(copy, "w").write(text.replace(m.group(1), m.group(1) + "\tv_mov_b64_e32 v[250:251], %s\n" % m.group(2), 1))
    res = cal.check_file(copy)
    assert any(any("compiler instruction" in w for w, _ in r["bad"]) for r in res.values())


def test_checker_follows_a_boolean_and_its_copy_through_a_branch():
    """Round 6 (k_median_banded, one-column form): the compiler tests `done` once (s_cselect_b64 -> s_and_b64 vcc, exec -> branch
    around the take-over and its wait) and re-tests a COPY of it in the block the exit path shares with the back edge.  The checker
    follows the boolean and its copies through the first branch, so the path `done -> no wait -> back edge` does not exist for it;
    a copy that is overwritten in between, or a take-over without its wait, must still be reported."""
    safe = """
\t;;#ASMSTART
\tglobal_load_dword v5, v[0:1], off
\t;;#ASMEND
.LBB0_1:
\ts_cmp_ge_i32 s4, s5
\ts_cselect_b64 s[2:3], -1, 0
\ts_mov_b64 s[18:19], s[2:3]
\ts_and_b64 vcc, exec, s[2:3]
\ts_cbranch_vccnz .LBB0_3
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\tv_mov_b32 v9, v5
\t;;#ASMEND
.LBB0_3:
\ts_andn2_b64 vcc, exec, s[18:19]
\ts_cbranch_vccz .LBB0_9
\tv_mov_b32_e32 v5, v1
\tv_add_u32_e32 v0, v0, v5
\t;;#ASMSTART
\tglobal_load_dword v5, v[0:1], off
\t;;#ASMEND
\ts_add_i32 s4, s4, 1
\ts_branch .LBB0_1
.LBB0_9:
\ts_endpgm
""".split("\n")
    assert cal.analyse(safe)["bad"] == []
    clobbered = list(safe)
    clobbered.insert(clobbered.index(".LBB0_3:") + 1, "\ts_mov_b64 s[18:19], s[6:7]")  # the copy no longer holds `done`: the back edge is reachable without the wait
    assert any("compiler instruction" in w for w, _ in cal.analyse(clobbered)["bad"])
    no_wait = [ln for ln in safe if "s_waitcnt" not in ln]
    assert cal.analyse(no_wait)["bad"]
