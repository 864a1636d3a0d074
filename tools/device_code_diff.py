#!/usr/bin/env python3
"""Is the device code of the working tree the same as at another commit?  Compiles the nine translation units of
adcensus_amd/csrc to gfx950 assembly with the product flags at both revisions (a temporary git worktree for the other one) and
compares them, ignoring comments, debug locations and the per-build `__hip_cuid_*` symbol.
    python tools/device_code_diff.py <commit>
(Used at the end of round 3: everything after the last GPU-validated commit, 13c55c7, changed tests, tools, documents and added
unused inline functions to adc_device_fn.h -- the generated code of all nine units is identical.)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = ["capi", "k_cost", "k_arms", "k_aggregate", "k_scanline", "k_wta", "k_refine", "k_voting", "k_paper"]


def isa(root, stem, out):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                    "-Wno-inline-asm", "-Wno-unused-value", "-Wno-unused-result", "--offload-device-only", "-S",
                    os.path.join(root, "adcensus_amd", "csrc", stem + ".hip"), "-o", out], capture_output=True, check=True)
    t = re.sub(r"__hip_cuid_[0-9a-f]+", "__hip_cuid_X", open(out).read())
    return "\n".join(l for l in t.split("\n") if not re.match(r"\s*(;|\.file|\.ident|\.loc)", l))


def functions(text):
    """{mangled name: body} of the functions of an assembly listing (label line up to its .Lfunc_end)."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        body = re.sub(r"\.L(BB|tmp|func_end|func_begin)[0-9_]+", ".L", m.group(2))  # (label numbers shift when functions are added)
        body = "\n".join(re.sub(r"\s*;.*$", "", l) for l in body.split("\n"))       # (... and so do the block numbers in trailing comments)
        out[m.group(1)] = body
    return out


def main():
    commit = sys.argv[1]
    tmp = tempfile.mkdtemp()
    other = os.path.join(tmp, "other")
    subprocess.run(["git", "-C", ROOT, "worktree", "add", "-q", other, commit], check=True)
    same = True
    try:
        for s in SRCS:
            a, b = isa(other, s, os.path.join(tmp, "a_%s.s" % s)), isa(ROOT, s, os.path.join(tmp, "b_%s.s" % s))
            if a == b:
                print("%-12s identical" % s)
                continue
            # per function: a unit that only GAINED kernels (new template instantiations) leaves the old ones untouched -- or not
            fa, fb = functions(a), functions(b)
            changed = [n for n in fa if n in fb and fa[n] != fb[n]]
            gone, added = [n for n in fa if n not in fb], [n for n in fb if n not in fa]
            print("%-12s %s: %d functions identical, %d changed, %d removed, %d new" % (s, "DIFFERENT" if changed or gone else "old functions identical",
                                                                                     len([n for n in fa if n in fb]) - len(changed), len(changed), len(gone), len(added)))
            for n in changed[:8]:
                print("             changed:", n[:110])
            for n in added[:8]:
                print("             new:    ", n[:110])
            same &= not changed and not gone
    finally:
        subprocess.run(["git", "-C", ROOT, "worktree", "remove", "--force", other])
        subprocess.run(["git", "-C", ROOT, "worktree", "prune"])
    print("device code of every function that exists at %s unchanged: %s" % (commit, same))
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
