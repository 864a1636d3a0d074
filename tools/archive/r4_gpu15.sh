#!/bin/bash
# round 4, GPU call 15 (experiment, not adopted in this round: no budget left for the full validation): K8 votes with the histogram
# cleared where it is read and a loop-free reduction for D <= 128 (-DIRV_VOTE_DIET) against the default build
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
L=ADC_HIP_LIB=$REPO/adcensus_amd/lib/k8diet/libadcensus_hip.so
env $L timeout 300 python -m pytest tests/test_gpu_stages.py -m gpu -x -q -k "stage_parity or middlebury" > $O/r4_gpu_pytest_k8diet.log 2>&1; echo "k8diet pytest rc=$? $(grep -E 'passed|failed' $O/r4_gpu_pytest_k8diet.log | tail -1)"
B="--no-cpu-baseline --no-extra-legs"
run() { TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4k_$TAG.json 2> $O/r4k_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4k_$TAG.json"))
    print("%-22s rc=$rc  %.1f pairs/s  %.3f ms  refine %.4f ms" % ("$TAG", d["value"], d["ms_per_step"], d["stage_ms"]["refine"]))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
for rep in 1 2; do
  run struct_default_$rep X=1 -- --workload structured --steps 10
  run struct_diet_$rep $L -- --workload structured --steps 10
done
run kitti_default X=1 -- --width 1242 --height 375 --workload structured --steps 30
run kitti_diet $L -- --width 1242 --height 375 --workload structured --steps 30
