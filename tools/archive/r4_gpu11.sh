#!/bin/bash
# round 4, GPU call 11: K6 marching kernel, every row cut into segments (do neighbouring streams help each other?)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
B="--no-cpu-baseline --no-extra-legs"
run() { TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4h_$TAG.json 2> $O/r4h_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4h_$TAG.json"))
    print("%-22s rc=$rc  %.1f pairs/s  %.3f ms  wta %.4f ms" % ("$TAG", d["value"], d["ms_per_step"], d["stage_ms"]["wta"]))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
run noise_plan X=1 -- --steps 20
for n in 2 3 4 6 8; do run noise_allrows_nseg$n ADC_WTA_NCU=100000 ADC_WTA_NSEG=$n -- --steps 20; done
run noise_plan_2 X=1 -- --steps 20
run noise_ncu128 ADC_WTA_NCU=128 -- --steps 20
run noise_ncu512 ADC_WTA_NCU=512 -- --steps 20
