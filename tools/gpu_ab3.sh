#!/bin/bash
# like gpu_ab.sh with REPS repetitions (default 3), compact output: usage  gpu_ab3.sh "ENV=a" "ENV=b" ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
WL=${WL:-noise}; REPS=${REPS:-3}
for rep in $(seq 1 $REPS); do
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --workload $WL 2>&1 | tail -1 | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); s = o['stage_ms']
print('%-28s pairs/s %.2f  agg %.3f scan %.3f wta %.3f refine %.3f  k4 %.4f' % ('$cfg', o['value'], s['aggregate'], s['scanline'], s['wta'], s['refine'], o['roofline']['avg_launch_ms']))"
done; done
