#!/bin/bash
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out/r5_12; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extra-legs"
for WL in noise structured; do
rm -rf "$REPO/$O/prof"
timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 $B --workload $WL > "$REPO/$O/rocprof.log" 2>&1; echo "rocprof rc=$?"
(cd "$REPO"; python tools/prof_summary.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | tail -1) | grep -i "interp\|itp\|kernel" | cut -c1-120)
done
rm -rf "$REPO/$O/prof"
