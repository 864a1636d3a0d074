#!/bin/bash
# rocprofv3 kernel stats of bench.py (in-flight 1) for WORKLOAD=${1:-noise}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; WL=${1:-noise}
cd /tmp && export TMPDIR=/tmp
rm -rf "$REPO/gpurun_out/prof_$WL"
timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_$WL" -o bench -- python "$REPO/bench.py" --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline --workload $WL > "$REPO/gpurun_out/rocprof_$WL.log" 2>&1; echo "rocprof rc=$?"
tail -1 "$REPO/gpurun_out/rocprof_$WL.log" | cut -c1-400
