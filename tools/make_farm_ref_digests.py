#!/usr/bin/env python3
"""Pins the batches bench.py measures on to the REFERENCE (test infrastructure; run in the build container).

For every pair of the two 1920x1080, D=128 bench batches -- noise seeds 12345 + i (BASELINE.json configs[3] / [4]) and
structured seeds 777 + i -- this runs the reference's own `ADCensusStereo::Initialize` + `Match`
(`ADCensusStereo.cpp:21-132`, through `oracle/_ref`'s `adc_oracle_match`, i.e. the reference sources compiled in place)
and writes the SHA-256 of the float32 disparity map into tests/golden/farm_ref_digests.json.

    make -C oracle ref && python tools/make_farm_ref_digests.py [--noise 20] [--structured 10] [--jobs 4]

About 55 s of one core and ~6 GB of host memory per pair.  Pairs already in the table are kept (the table only grows);
`--redo` recomputes everything.  bench.py's `farm_check.reference_*` fields and tests/test_gpu_fullsize.py compare the
HIP outputs with THIS table (the older tests/golden/farm_selfcheck_digests.json holds 1-GPU outputs of the product
itself: a cross-box repeatability table, not a parity claim).
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H, D = 1920, 1080, 128
OUT = os.path.join(ROOT, "tests", "golden", "farm_ref_digests.json")


def one(job):
    workload, pid = job
    from adcensus_amd import workloads
    from oracle import pyoracle
    ref = pyoracle.load("reference")
    assert ref.kind == "reference", ref.kind
    left, right = (workloads.noise_pair(W, H, 12345 + pid) if workload == "noise"
                   else workloads.structured_pair(W, H, D, seed=777 + pid))
    t = time.time()
    disp, secs = ref.match(left, right, pyoracle.Option(max_disparity=D))
    return workload, pid, hashlib.sha256(disp.tobytes()).hexdigest(), secs, time.time() - t, ref.build_info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--noise", type=int, default=20)
    ap.add_argument("--structured", type=int, default=10)
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--redo", action="store_true")
    a = ap.parse_args()
    table = {"_generator": "tools/make_farm_ref_digests.py", "_oracle": "oracle/_ref = reference sources compiled in place (adc_oracle_match = ADCensusStereo.cpp:21-132)",
             "size": [W, H, D], "seeds": {"noise": "12345 + pair id", "structured": "777 + pair id"}, "noise": {}, "structured": {}}
    if os.path.exists(OUT) and not a.redo:
        with open(OUT) as f:
            old = json.load(f)
        if old.get("size") == [W, H, D]:
            table["noise"].update(old.get("noise", {}))
            table["structured"].update(old.get("structured", {}))
    jobs = [("noise", i) for i in range(a.noise) if str(i) not in table["noise"]]
    jobs += [("structured", i) for i in range(a.structured) if str(i) not in table["structured"]]
    print("%d pairs to run with %d processes" % (len(jobs), a.jobs), flush=True)
    with mp.get_context("spawn").Pool(a.jobs) as pool:
        for workload, pid, dig, secs, wall, info in pool.imap_unordered(one, jobs):
            table[workload][str(pid)] = dig
            table["_build_info"] = info
            print("%-10s pair %2d  %s  Match %.1f s (wall %.1f)" % (workload, pid, dig[:16], secs, wall), flush=True)
            with open(OUT + ".tmp", "w") as f:
                json.dump(table, f, indent=1, sort_keys=True)
            os.replace(OUT + ".tmp", OUT)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
