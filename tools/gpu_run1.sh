#!/bin/bash
# first GPU session: parity diagnostics, timings, pytest -m gpu, bench, rocprof kernel stats
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/diag.jsonl
export PYTHONUNBUFFERED=1
timeout 900 python tools/gpu_diag.py parity > gpurun_out/diag_parity.log 2>&1; echo "parity rc=$?"
timeout 600 python tools/gpu_diag.py timing > gpurun_out/diag_timing.log 2>&1; echo "timing rc=$?"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 6 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -2 gpurun_out/bench.log
REPO="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_r1" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > "$REPO/gpurun_out/rocprof.log" 2>&1; echo "rocprof rc=$?"
ls -R "$REPO/gpurun_out/prof_r1" | head -20
