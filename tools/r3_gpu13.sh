#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_stages.py "tests/test_gpu_fullsize.py::test_full_size_match_equals_reference" "tests/test_gpu_fullsize.py::test_kitti_size_match_equals_reference" -m gpu -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT/$O/prof_g13"
timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_g13" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > "$GRAFT_REPO_ROOT/$O/rocprof_g13.log" 2>&1
cd "$GRAFT_REPO_ROOT"; python tools/prof_summary.py $(ls $O/prof_g13/*.db $O/prof_g13/*/*.db 2>/dev/null | tail -1) > $O/g13_stats.md; grep -i "wta\|median" $O/g13_stats.md
