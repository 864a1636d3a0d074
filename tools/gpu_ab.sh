#!/bin/bash
# same-box A/B of environment switches: usage  gpu_ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument = one configuration)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
WL=${WL:-noise}
for rep in 1 2; do
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --workload $WL 2>&1 | tail -1 | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('   pairs/s %.2f  ms %.3f  stages %s  k4 %.4f ms' % (o['value'], o['ms_per_step'], o['stage_ms'], o['roofline']['avg_launch_ms']))"
done; done
