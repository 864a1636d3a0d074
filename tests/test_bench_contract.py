"""bench.py's one-line JSON contract, checked on the committed sample outputs (profiles/r1_bench_*.json were written
by `python bench.py` on an MI355X) and on the helpers that do not need a GPU."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


@pytest.mark.parametrize("name", ["r1_bench_default.json", "r1_bench_structured.json"])
def test_committed_bench_lines_follow_the_contract(name):
    path = os.path.join(ROOT, "profiles", name)
    lines = [l for l in open(path).read().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    o = json.loads(lines[0])
    for k, t in REQUIRED.items():
        assert k in o and isinstance(o[k], t), k
    assert "vs_baseline" in o and o["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert o["unit"] == "pairs/s" and o["higher_is_better"] is True and o["scaling"] == "weak" and o["dtype"] == "f32"
    assert "workload" in o["config"] and "model" not in o["config"]
    r = o["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(o["value"] - 1000.0 / o["ms_per_step"] * o["n_gpus"]) / o["value"] < 0.02  # one object in flight, N = 1
    if "cpu_baseline" in o:
        c = o["cpu_baseline"]
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("reference", "port") and c["cores"] == 1


def test_pmc_traffic_helper_reads_the_committed_profiles():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for wl in ("noise", "structured"):
        t = bench.pmc_traffic(wl, (1920, 1080, 128))
        assert t is not None and 2.1e9 < t < 2.4e9, (wl, t)  # ~2.13 GB algorithmic + segment halos
    assert bench.pmc_traffic("noise", (640, 480, 64)) is None
