#!/bin/bash
# round 4, GPU call 13: K8 votes, the pixels of a block's first (and second) bin in ONE LDS atomic each (IRV_MERGE_BINS 0 = before,
# 1 = first bin, 2 = first two bins = the default build): parity, then same-box A/B on the structured pairs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_api.py -m gpu -x -q -k "stage_parity or middlebury or voting or budget" > $O/r4_gpu_pytest_k8merge.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/r4_gpu_pytest_k8merge.log | tail -2
grep -q " passed" $O/r4_gpu_pytest_k8merge.log && ! grep -q "failed" $O/r4_gpu_pytest_k8merge.log || { echo "TESTS NOT GREEN -- stopping"; exit 1; }
B="--no-cpu-baseline --no-extra-legs"
run() { TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4i_$TAG.json 2> $O/r4i_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4i_$TAG.json"))
    print("%-22s rc=$rc  %.1f pairs/s  %.3f ms  refine %.4f ms" % ("$TAG", d["value"], d["ms_per_step"], d["stage_ms"]["refine"]))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
for rep in 1 2; do
  run struct_merge0_$rep ADC_HIP_LIB=$REPO/adcensus_amd/lib/k8merge0/libadcensus_hip.so -- --workload structured --steps 10
  run struct_merge1_$rep ADC_HIP_LIB=$REPO/adcensus_amd/lib/k8merge1/libadcensus_hip.so -- --workload structured --steps 10
  run struct_merge2_$rep X=1 -- --workload structured --steps 10
done
run kitti_merge0 ADC_HIP_LIB=$REPO/adcensus_amd/lib/k8merge0/libadcensus_hip.so -- --width 1242 --height 375 --workload structured --steps 30
run kitti_merge1 ADC_HIP_LIB=$REPO/adcensus_amd/lib/k8merge1/libadcensus_hip.so -- --width 1242 --height 375 --workload structured --steps 30
run kitti_merge2 X=1 -- --width 1242 --height 375 --workload structured --steps 30
