#!/bin/bash
# parity tests + single-object timings + bench (+ optional A/B env)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/diag.jsonl
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log
timeout 600 python tools/gpu_diag.py timing > gpurun_out/diag_timing.log 2>&1; echo "timing rc=$?"
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --workload structured > gpurun_out/bench_struct.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench_struct.log
REPO="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --workload structured > "$REPO/gpurun_out/rocprof.log" 2>&1; echo "rocprof rc=$?"
