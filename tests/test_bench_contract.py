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
    if not name.startswith("r3_"):
        assert o["scaling"] == "weak"
    assert "workload" in o["config"] and "model" not in o["config"]
    _check_roofline(o["roofline"])
    if o["config"].get("in_flight_per_gpu", 1) == 1:
        assert abs(o["value"] - 1000.0 / o["ms_per_step"] * o["n_gpus"]) / o["value"] < 0.02
    if "cpu_baseline" in o:
        c = o["cpu_baseline"]
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("reference", "port") and c["cores"] == 1
    if name.startswith("r3_"):  # round-3 lines: frac = REAL HBM bytes / time / peak (<= 1), the SURVEY 8d figure is algorithmic_frac
        r = o["roofline"]
        assert r["frac"] <= 1.0 and r["algorithmic_frac"] >= r["frac"] - 1e-9 and r["passes_per_launch"] >= 1.0
        assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
        assert o["farm_check"]["ok"] is True and o["farm_check"]["done_counter"] == (o["steps"] if o["scaling"] == "strong" else o["steps"] * o["n_gpus"])
        assert "committed_1gpu_checked" in o["farm_check"] and "reference_checked" not in o["farm_check"]
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


def test_round2_sample_is_committed():
    assert any(n.startswith("r2_bench_") for n in SAMPLES), "commit a round-2 bench.py output under profiles/"


def test_pmc_traffic_helper():
    bench = _bench()
    assert bench.pmc_traffic("noise", (640, 480, 64)) is None
    assert len(bench.k4_source_hash()) == 16
    for wl in ("noise", "structured"):
        p = os.path.join(ROOT, "profiles", "r3_k4_pmc_traffic_%s.json" % wl)
        t = bench.pmc_traffic(wl, (1920, 1080, 128))
        if os.path.exists(p) and json.load(open(p)).get("k4_src_sha16") == bench.k4_source_hash():
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
