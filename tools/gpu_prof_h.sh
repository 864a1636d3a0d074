#!/bin/bash
# rocprofv3 kernel stats of bench.py at a given image height (scaling experiment): gpu_prof_h.sh HEIGHT
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; HT=${1:-540}
cd /tmp && export TMPDIR=/tmp
rm -rf "$REPO/gpurun_out/prof_h$HT"
timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_h$HT" -o bench -- python "$REPO/bench.py" --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline --height $HT > "$REPO/gpurun_out/rocprof_h$HT.log" 2>&1; echo "rocprof rc=$?"
