#!/bin/bash
# round 4, GPU call 16: launch shape of the voting chain after the cheaper votes (ADC_IRV_GRID / ADC_IRV_WPB), structured pairs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
B="--no-cpu-baseline --no-extra-legs"
run() { TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4l_$TAG.json 2> $O/r4l_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4l_$TAG.json"))
    print("%-22s rc=$rc  %.1f pairs/s  %.3f ms  refine %.4f ms" % ("$TAG", d["value"], d["ms_per_step"], d["stage_ms"]["refine"]))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
run s_default X=1 -- --workload structured --steps 10
run s_grid256 ADC_IRV_GRID=256 -- --workload structured --steps 10
run s_grid1024 ADC_IRV_GRID=1024 -- --workload structured --steps 10
run s_wpb8 ADC_IRV_WPB=8 -- --workload structured --steps 10
run s_default2 X=1 -- --workload structured --steps 10
run k_default X=1 -- --width 1242 --height 375 --workload structured --steps 30
run k_grid128 ADC_IRV_GRID=128 -- --width 1242 --height 375 --workload structured --steps 30
run k_grid512 ADC_IRV_GRID=512 -- --width 1242 --height 375 --workload structured --steps 30
