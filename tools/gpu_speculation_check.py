"""Cross-check of the two speculative forms of round 4 against their plain forms on many DIFFERENT pairs (tools/, not a test:
~1 GPU-minute): the median's speculative bands (ADC_MEDIAN_SPEC) and the scanline row passes cut into verified segments
(ADC_SO_SEG) are switched per process, so this script runs itself twice -- default switches, then ADC_MEDIAN_SPEC=0 ADC_SO_SEG=0 --
over the same seeded pairs (1080p noise, 1080p structured, KITTI-size structured, Middlebury-size structured) and compares the
SHA-256 of every output map; it also prints how often a seam failed and was redone (which is allowed: the result must be the same).
  python tools/gpu_speculation_check.py [pairs per workload]"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.getcwd())


def worker(n):
    import adcensus_amd as A
    from adcensus_amd import workloads
    out = {}
    for tag, W, H, D, gen in (("noise1080", 1920, 1080, 128, lambda s: workloads.noise_pair(1920, 1080, 5000 + s)),
                              ("struct1080", 1920, 1080, 128, lambda s: workloads.structured_pair(1920, 1080, 128, seed=6000 + s)),
                              ("structkitti", 1242, 375, 128, lambda s: workloads.structured_pair(1242, 375, 128, seed=7000 + s)),
                              ("struct450", 450, 375, 64, lambda s: workloads.structured_pair(450, 375, 64, seed=8000 + s))):
        st = A.ADCensusStereo(device=0)
        assert st.Initialize(W, H, A.ADCensusOption(min_disparity=0, max_disparity=D))
        k = n if "1080" not in tag else max(2, n // 2)
        for s in range(k):
            l, r = gen(s)
            out["%s/%d" % (tag, s)] = hashlib.sha256(st.match(l, r).tobytes()).hexdigest()
        out["%s/counters" % tag] = {"median_fallbacks": st.debug_counter(0), "median_seam_failures": st.debug_counter(7),
                                    "scanline_seam_redos": st.debug_counter(4), "segments_per_row": st.debug_counter(5),
                                    "median_speculative": st.debug_counter(8)}
        st.Release()
    print(json.dumps(out))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        return worker(int(sys.argv[2]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    res = []
    for env in ({}, {"ADC_MEDIAN_SPEC": "0", "ADC_SO_SEG": "0"}):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(n)], env=dict(os.environ, **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    a, b = res
    keys = [k for k in a if not k.endswith("/counters")]
    bad = [k for k in keys if a[k] != b[k]]
    print("pairs compared: %d, outputs that differ between the speculative and the plain forms: %d %s" % (len(keys), len(bad), bad[:5]))
    for k in a:
        if k.endswith("/counters"):
            print("  %-22s speculative forms: %s | plain forms: %s" % (k, a[k], b[k]))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
