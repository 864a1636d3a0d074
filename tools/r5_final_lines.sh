#!/bin/bash
# round 5: the bench lines of the final tree once more with the PMC files of tools/r5_final.sh in place (roofline.traffic)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; export PYTHONUNBUFFERED=1
timeout 400 python bench.py > $O/r5_bench_default.json 2> $O/r5_bench_default.err; echo "default rc=$?"; cut -c1-200 $O/r5_bench_default.json
timeout 400 python bench.py --workload structured --steps 10 --cpu-baseline-structured > $O/r5_bench_structured.json 2> $O/r5_bench_structured.err; echo "structured rc=$?"
B="--no-cpu-baseline --no-extra-legs"
for WL in noise structured; do
  timeout 100 python bench.py --width 1242 --height 375 --workload $WL --steps 30 $B > $O/r5_bench_kitti_$WL.json 2> $O/r5_bench_kitti_$WL.err; echo "kitti $WL rc=$?"
done
python - <<'P'
import json
for n in ("default", "structured", "kitti_noise", "kitti_structured"):
    o = json.load(open("gpurun_out/r5_bench_%s.json" % n)); r = o["roofline"]
    print(n, o["value"], "frac", r["frac"], "traffic", r["traffic"], r.get("traffic_over_bytes"), o["farm_check"]["ok"])
P
