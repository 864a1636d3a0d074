#!/bin/bash
# round 4: why do the boxes differ?  Shader / memory clock and power (rocm-smi, one sample per second) while bench.py runs the headline
# workload for a few seconds, next to the line it prints.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-a}; O=gpurun_out
timeout 200 python bench.py --no-cpu-baseline --no-extra-legs --steps 1500 > $O/r4_clk_$TAG.json 2> $O/r4_clk_$TAG.err &
BP=$!
: > $O/r4_clk_$TAG.txt
for i in $(seq 1 40); do
  kill -0 $BP 2>/dev/null || break
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|socclk|Power' | sed -e 's/^GPU\[0\][ \t]*: //' | tr '\n' ';')" >> $O/r4_clk_$TAG.txt
  sleep 1
done
wait $BP
python - <<PY
import json
d = json.load(open("$O/r4_clk_$TAG.json")); r = d["roofline"]
print("%.1f pairs/s %.3f ms  K4 %.4f ms frac %.3f  copy %.0f GB/s  stages %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["device_copy_GBps"], d["stage_ms"]))
PY
cat $O/r4_clk_$TAG.txt | cut -c1-260 | tail -16
