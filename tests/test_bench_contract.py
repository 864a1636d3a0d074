"""bench.py's one-line JSON contract, checked on the committed sample outputs (profiles/r*_bench_*.json were written by
`python bench.py` on an MI355X) and on the helpers that do not need a GPU."""
import glob
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}
SAMPLES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*.json")))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _check_roofline(r):
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


@pytest.mark.parametrize("name", SAMPLES)
def test_committed_bench_lines_follow_the_contract(name):
    path = os.path.join(ROOT, "profiles", name)
    lines = [l for l in open(path).read().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    o = json.loads(lines[0])
    for k, t in REQUIRED.items():
        assert k in o and isinstance(o[k], t), k
    assert "vs_baseline" in o and o["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert o["unit"] == "pairs/s" and o["higher_is_better"] is True and o["scaling"] in ("weak", "strong") and o["dtype"] == "f32"
    if not name.startswith(("r3_", "r4_", "r5_", "r6_")):
        assert o["scaling"] == "weak"
    assert "workload" in o["config"] and "model" not in o["config"]
    _check_roofline(o["roofline"])
    if o["config"].get("in_flight_per_gpu", 1) == 1:
        assert abs(o["value"] - 1000.0 / o["ms_per_step"] * o["n_gpus"]) / o["value"] < 0.02
    if "cpu_baseline" in o:
        c = o["cpu_baseline"]
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("reference", "port") and c["cores"] == 1
    if name.startswith(("r5_", "r6_")):  # round 5 on: the batch is pinned to the REFERENCE CPU program's digests
        fc = o["farm_check"]
        assert fc["ok"] is True and fc["reference_mismatches"] == [] and "committed_1gpu_checked" not in fc
        if o["config"]["workload"].startswith(("noise 1920x1080", "structured 1920x1080")):
            assert fc["reference_checked"] == min(fc["pairs"], 20 if o["config"]["workload"].startswith("noise") else 10)
        for leg in ("structured", "noise", "mixed_stream"):
            if leg in o and "reference_check" in o[leg]:
                assert o[leg]["reference_check"]["reference_mismatches"] == []
        if "mixed_stream" in o:
            assert o["mixed_stream"]["reference_mismatches"] == [] and o["mixed_stream"]["reference_checked"] > 0
        r = o["roofline"]
        assert r["frac"] <= 1.0 and abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    if name.startswith("r6_") and "match_host" in o:  # round 6: the drop-in figure and the natural-image legs are part of the line
        mh = o["match_host"]
        wl = o["config"]["workload"].split()[0]
        other = "structured" if wl == "noise" else "noise"
        assert mh[wl]["pageable"]["value"] > 0 and mh[wl]["registered"]["value"] > 0 and mh[other]["pageable"]["value"] > 0
        assert mh[wl]["pageable"]["value"] < o["value"] * 1.05  # the host path cannot beat the resident one
        assert o[other]["throughput_mode"]["value"] > 0 and o[other]["host_inclusive"]["value"] > 0 and "voting" in o[other]
        if "cpu_baseline_all_cores" in o:
            a = o["cpu_baseline_all_cores"]
            assert a["processes"] == a["cores"] >= 2 and a["value"] > o["cpu_baseline"]["value"]
    if name.startswith(("r3_", "r4_")):  # lines of rounds 3 and 4: frac = REAL HBM bytes / time / peak (<= 1), the SURVEY 8d figure is algorithmic_frac
        r = o["roofline"]
        assert r["frac"] <= 1.0 and r["algorithmic_frac"] >= r["frac"] - 1e-9 and r["passes_per_launch"] >= 1.0
        assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
        assert o["farm_check"]["ok"] is True and o["farm_check"]["done_counter"] == (o["steps"] if o["scaling"] == "strong" else o["steps"] * o["n_gpus"])
        assert "committed_1gpu_checked" in o["farm_check"] and "reference_checked" not in o["farm_check"]  # (rounds 3 / 4: the product's own table)
        for k, v in o["stage_roofline"].items():
            if isinstance(v, dict) and k != "aggregate_K4_stage_algorithmic":
                assert v["frac"] <= 1.0, (k, v)
        for leg in ("structured", "noise"):
            if leg in o:
                _check_roofline(o[leg]["roofline"])
                assert o[leg]["roofline"]["frac"] <= 1.0
        if o["n_gpus"] > 1:
            assert o["config"]["comm"]["world_size"] == o["n_gpus"] == o["config"]["comm"]["ranks_in_all_reduce"]
    if name.startswith("r2_"):  # round-2 lines: the farm check and the extra legs
        assert o["farm_check"]["ok"] is True and o["farm_check"]["done_counter"] == o["steps"] * o["n_gpus"]
        r = o["roofline"]
        assert r["hbm_frac"] <= r["frac"] + 1e-9 and r["passes_per_launch"] >= 1.0
        for leg in ("structured", "noise"):
            if leg in o:
                _check_roofline(o[leg]["roofline"])
        if o["n_gpus"] == 1 and "host_inclusive" in o:
            assert o["host_inclusive"]["value"] < o["value"] * 1.05  # the host path cannot beat the resident one
            assert set(("cpu_model", "build", "host_cores")) <= set(o["cpu_baseline"])


def test_reference_digest_table_of_the_bench_batches():
    """tests/golden/farm_ref_digests.json (reference CPU program, tools/make_farm_ref_digests.py) covers the 20 noise pairs of the
    headline batch and 10 structured pairs; where it overlaps with the older table of the product's own 1-GPU outputs
    (farm_selfcheck_digests.json, noise pairs 0..159) the two agree -- the product's outputs ARE the reference's."""
    gold = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gold, "farm_ref_digests.json")) as f:
        ref = json.load(f)
    assert ref["size"] == [1920, 1080, 128] and "oracle/_ref" in ref["_oracle"]
    assert sorted(int(k) for k in ref["noise"]) == list(range(20)) and sorted(int(k) for k in ref["structured"]) == list(range(10))
    assert all(len(v) == 64 and int(v, 16) >= 0 for w in ("noise", "structured") for v in ref[w].values())
    with open(os.path.join(gold, "farm_selfcheck_digests.json")) as f:
        own = json.load(f)
    assert own["workload"] == "noise" and own["size"] == ref["size"]
    assert all(own["digests"][k] == v for k, v in ref["noise"].items())
    from adcensus_amd import farm
    r, s_ = farm.load_digest_tables(ROOT, "noise", [1920, 1080, 128])
    assert r == ref["noise"] and s_ == own["digests"]
    assert farm.load_digest_tables(ROOT, "structured", [1920, 1080, 128]) == (ref["structured"], None)
    assert farm.load_digest_tables(ROOT, "noise", [640, 480, 64]) == (None, None)


def test_round2_sample_is_committed():
    assert any(n.startswith("r2_bench_") for n in SAMPLES), "commit a round-2 bench.py output under profiles/"


def test_pmc_traffic_helper():
    bench = _bench()
    assert bench.pmc_traffic("noise", (640, 480, 64)) is None
    assert len(bench.k4_source_hash()) == 16
    for wl in ("noise", "structured"):
        ps = [os.path.join(ROOT, "profiles", "r%d_k4_pmc_traffic_%s.json" % (rnd, wl)) for rnd in (6, 5, 4, 3)]  # (bench.py's order)
        t = bench.pmc_traffic(wl, (1920, 1080, 128))
        if any(os.path.exists(p) and json.load(open(p)).get("k4_src_sha16") == bench.k4_source_hash() for p in ps):
            assert t is not None and 2.1e9 < t < 2.4e9, (wl, t)  # ~2.13 GB algorithmic + segment halos
        else:
            assert t is None  # a measurement on other kernel sources must not be reported


def test_roofline_arithmetic():
    """k4_roofline on synthetic aggregate_info tuples: frac is the REAL traffic figure (a pair launch reads one volume and
    writes one), algorithmic_frac counts two passes of SURVEY-8d bytes for it."""
    bench = _bench()

    class NoLib:
        def adc_device_malloc(self, n):
            return 0

        def adc_device_free(self, p):
            pass
    W, H, D = 1920, 1080, 128
    P, V = float(W * H), 4.0 * W * H * D
    r = bench.k4_roofline([(0.5, 7, 7, True)], W, H, D, NoLib(), "structured")  # 7 single launches after the fused first pass
    assert abs(r["algorithmic_bytes_per_launch"] - (7 * (2 * V + 4 * P) + 8 * P) / 7) < 1 and r["passes_per_launch"] == 1.0
    assert abs(r["achieved"] - r["algorithmic_achieved"]) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    r = bench.k4_roofline([(0.5, 4, 7, True)], W, H, D, NoLib(), "noise")  # 3 pairs + the last pass
    assert abs(r["algorithmic_bytes_per_launch"] - (7 * (2 * V + 4 * P) + 8 * P) / 4) < 1
    assert abs(r["bytes_per_launch"] - (4 * (2 * V + 4 * P) + 8 * P) / 4) < 1
    assert r["algorithmic_frac"] > r["frac"] and r["frac"] <= 1.0 and abs(r["passes_per_launch"] - 1.75) < 1e-9
    sr = bench.stage_roofline({"scanline": 2.0, "wta": 0.3, "aggregate": 3.5}, 6.0, W, H, D)
    assert abs(sr["scanline_K5"]["bytes"] - 4 * (2 * V + 3 * P)) < 1 and abs(sr["wta_right_K6"]["bytes"] - V) < 1
    assert abs(sr["whole_match"]["bytes"] - 26 * V) < 1 and all(v["frac"] < 1.0 for v in sr.values())


def test_bench_two_ranks_gloo_stub():
    """First-contact insurance for the multi-GPU path (round-3 review item 8): `bench.py --gpus 2` end to end on this machine --
    the torchrun re-exec, one process per rank, LOCAL_RANK -> device binding, torch.distributed collectives (gloo here, RCCL on
    the node), static partition + neighbour re-check, the host-fed leg, the scaling reference and the ONE JSON line of rank 0 --
    with a stub matcher (tests/bench_stub.py) in place of the HIP library.  The first real 8-GPU run is then not the first time
    this code runs with more than one rank."""
    import subprocess
    import sys
    env = dict(os.environ, ADC_BENCH_BACKEND="gloo", ADC_BENCH_MATCHER_MODULE="tests.bench_stub", PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "64", "--height", "48",
                        "--disp", "16", "--spinup-ms", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout: rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["comm"]["ranks_in_all_reduce"] == 2 and d["config"]["comm"]["world_size"] == 2
    assert d["farm_check"]["ok"], d["farm_check"]
    assert d["device_binding"]["device_index"] == d["device_binding"]["local_rank_env"] == 0  # rank 0 -> device LOCAL_RANK
    assert d["device_binding"]["matcher_module"] == "tests.bench_stub"
    assert "scaling_reference" in d and d["scaling_reference"]["n_gpus"] == 1
    assert "host_farm" in d and d["host_farm"]["n_gpus"] == 2
    assert d["value"] > 0 and d["unit"] == "pairs/s" and d["higher_is_better"] is True


def test_bench_pull_queue_two_ranks_gloo_stub():
    """The same for the fixed batch through the pull queue (BASELINE.json configs[4]: --batch B, strong scaling)."""
    import subprocess
    import sys
    env = dict(os.environ, ADC_BENCH_BACKEND="gloo", ADC_BENCH_MATCHER_MODULE="tests.bench_stub", PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--warmup", "1", "--width", "64", "--height", "48",
                        "--disp", "16", "--spinup-ms", "0", "--no-host-leg"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["farm_check"]["ok"]
    assert d["config"]["comm"]["ranks_in_all_reduce"] == 2


def test_bench_exits_nonzero_when_rccl_was_asked_for_and_gloo_carried_the_job():
    """Round-5 review: on the first real multi-GPU run a broken RCCL must not look like a pass.  Here (no GPU) the RCCL communicator
    cannot be created: the job still runs over gloo and prints its line -- labelled -- but the exit code is 3 (torchrun reports
    the failing ranks; the re-exec'ing parent hands torchrun's code on)."""
    import subprocess
    import sys
    env = dict(os.environ, ADC_BENCH_MATCHER_MODULE="tests.bench_stub", PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "RANK", "ADC_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "64", "--height", "48",
                        "--disp", "16", "--spinup-ms", "0", "--no-host-leg"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["config"]["comm_backend"] == "gloo (RCCL failed)" and d["farm_check"]["ok"]
