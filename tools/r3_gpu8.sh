#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_api.py "tests/test_gpu_fullsize.py::test_full_size_match_equals_reference" "tests/test_gpu_fullsize.py::test_kitti_size_match_equals_reference" -m gpu -x -q 2>&1 | tail -6
for WL in noise structured; do
timeout 600 python bench.py --workload $WL --steps 10 --no-cpu-baseline --no-extra-legs > $O/g8_$WL.json 2> $O/g8_$WL.err
python - $WL <<'PY'
import json, sys
o = json.loads(open("gpurun_out/g8_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value", o["value"], o["stage_ms"], o["roofline"]["avg_launch_ms"], o["roofline"]["frac"], o["async_fallbacks"])
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT/$O/prof_g8"
timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_g8" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > "$GRAFT_REPO_ROOT/$O/rocprof_g8.log" 2>&1
cd "$GRAFT_REPO_ROOT"; python tools/prof_summary.py $(ls $O/prof_g8/*.db $O/prof_g8/*/*.db 2>/dev/null | tail -1) | head -12
