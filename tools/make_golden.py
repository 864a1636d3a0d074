#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the REAL reference build
(oracle/_ref, i.e. the reference's own sources compiled in place with the canonical recipe).

Run in the build container (where /root/reference exists):
    make -C oracle ref && python tools/make_golden.py

Writes
  tests/golden/cone_pair.npz     the Middlebury Cone pair of the reference's Data/ dir as BGR arrays
                                 (BASELINE.json configs[0]/[1] input; PNG decode is lossless)
  tests/golden/golden.json       SHA-256 of every stage dump of the reference for a list of named,
                                 seeded cases (Cone + small synthetic cases + option variants)
  tests/golden/cone_final.npy.gz not written: the final map is pinned by its SHA-256 only
  tests/golden/{cloth3,piano,wood2}_pair.npz  the other pairs of the reference's Data/ dir (committed too)
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from tests import cases  # noqa: E402

REF_DATA = "/root/reference/Data"


def bgr(path):
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGB"))[:, :, ::-1])


def main():
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    if os.path.isdir(REF_DATA):
        np.savez_compressed(os.path.join(gold, "cone_pair.npz"),
                            left=bgr(os.path.join(REF_DATA, "Cone", "im2.png")),
                            right=bgr(os.path.join(REF_DATA, "Cone", "im6.png")))
        data = gold  # committed (about 4.7 MB): the GPU parity tests of these pairs must run on a fresh clone
        for name, l, r in (("cloth3", "Cloth3/view1.png", "Cloth3/view5.png"), ("piano", "Piano/im0.png", "Piano/im1.png"),
                           ("wood2", "Wood2/view1.png", "Wood2/view5.png")):
            np.savez_compressed(os.path.join(data, name + "_pair.npz"), left=bgr(os.path.join(REF_DATA, l)),
                                right=bgr(os.path.join(REF_DATA, r)))
    ref = pyoracle.load("reference")
    out = {"_generator": "tools/make_golden.py", "_oracle": "oracle/_ref (reference sources, canonical recipe)", "cases": {}}
    for name in cases.GOLDEN_CASES:
        left, right, opt = cases.make_case(name)
        dumps = ref.run(left, right, opt)
        out["cases"][name] = {k: hashlib.sha256(cases.canonical(k, v, opt).tobytes()).hexdigest() for k, v in dumps.items()}
        print(name, left.shape, "final", out["cases"][name]["disp_final"][:16], flush=True)
    with open(os.path.join(gold, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
