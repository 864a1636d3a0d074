#!/bin/bash
# round 5: the headline and the structured line of the final tree on one more box (spread of the pool / clock states)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; export PYTHONUNBUFFERED=1
B="--no-cpu-baseline --no-extra-legs --steps 20"
for rep in 1 2; do
for WL in noise structured; do
  timeout 200 python bench.py $B --workload $WL > $O/b.json 2>/dev/null
  python - "$WL" <<'P' | tee -a $O/r5_box_spread.txt
import json, sys
o = json.load(open('gpurun_out/b.json'))
print(sys.argv[1], "pairs/s %.1f" % o['value'], "stage ms", {k: round(v, 3) for k, v in o['stage_ms'].items()}, "K4 launch %.4f ms frac %.3f" % (o['roofline']['avg_launch_ms'], o['roofline']['frac']), "reference digests", o['farm_check']['reference_checked'], o['farm_check']['reference_mismatches'])
P
done
done
