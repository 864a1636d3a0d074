#!/bin/bash
# PMC passes (separate runs per counter group, kernel-trace only) for the aggregation kernel's HBM traffic
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; WL=${1:-noise}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$REPO/gpurun_out/pmc_${WL}_$C"
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc_${WL}_$C" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --workload $WL > "$REPO/gpurun_out/pmc_${WL}_$C.log" 2>&1; echo "$C rc=$?"
  ls "$REPO/gpurun_out/pmc_${WL}_$C" | head
done
