#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_stages.py "tests/test_gpu_api.py::test_median_handoff_timeout_falls_back" "tests/test_gpu_fullsize.py::test_full_size_match_equals_reference" "tests/test_gpu_fullsize.py::test_kitti_size_match_equals_reference" -m gpu -x -q 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
for MP in 1 0; do
rm -rf "$GRAFT_REPO_ROOT/$O/prof_g9_$MP"
ADC_MEDIAN_PAIRS=$MP timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_g9_$MP" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > "$GRAFT_REPO_ROOT/$O/rocprof_g9_$MP.log" 2>&1
(cd "$GRAFT_REPO_ROOT"; echo "ADC_MEDIAN_PAIRS=$MP"; python tools/prof_summary.py $(ls $O/prof_g9_$MP/*.db $O/prof_g9_$MP/*/*.db 2>/dev/null | tail -1) > $O/g9_stats_$MP.md; grep -i "median\|interpolate_tab" $O/g9_stats_$MP.md)
done
cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-extra-legs > $O/g9_noise.json 2> $O/g9_noise.err
python - <<'PY'
import json
o = json.loads(open("gpurun_out/g9_noise.json").read().strip().splitlines()[-1])
print("noise value", o["value"], o["stage_ms"], o["async_fallbacks"])
PY
