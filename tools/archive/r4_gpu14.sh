#!/bin/bash
# round 4, GPU call 14: K8 votes, three bins merged (then pixel by pixel) and one atomic per DISTINCT bin (IRV_MERGE_BINS 3 / 8) against 2
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
for v in k8merge3 k8merge8; do
  env ADC_HIP_LIB=$REPO/adcensus_amd/lib/$v/libadcensus_hip.so timeout 300 python -m pytest tests/test_gpu_stages.py -m gpu -x -q -k "stage_parity or middlebury" > $O/r4_gpu_pytest_$v.log 2>&1; echo "$v pytest rc=$? $(grep -E 'passed|failed' $O/r4_gpu_pytest_$v.log | tail -1)"
done
B="--no-cpu-baseline --no-extra-legs"
run() { TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4j_$TAG.json 2> $O/r4j_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4j_$TAG.json"))
    print("%-22s rc=$rc  %.1f pairs/s  %.3f ms  refine %.4f ms" % ("$TAG", d["value"], d["ms_per_step"], d["stage_ms"]["refine"]))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
for rep in 1 2; do
  run struct_merge2_$rep X=1 -- --workload structured --steps 10
  run struct_merge3_$rep ADC_HIP_LIB=$REPO/adcensus_amd/lib/k8merge3/libadcensus_hip.so -- --workload structured --steps 10
  run struct_merge8_$rep ADC_HIP_LIB=$REPO/adcensus_amd/lib/k8merge8/libadcensus_hip.so -- --workload structured --steps 10
done
run kitti_merge2 X=1 -- --width 1242 --height 375 --workload structured --steps 30
run kitti_merge3 ADC_HIP_LIB=$REPO/adcensus_amd/lib/k8merge3/libadcensus_hip.so -- --width 1242 --height 375 --workload structured --steps 30
run kitti_merge8 ADC_HIP_LIB=$REPO/adcensus_amd/lib/k8merge8/libadcensus_hip.so -- --width 1242 --height 375 --workload structured --steps 30
run noise_merge2 X=1 -- --steps 20
run noise_merge8 ADC_HIP_LIB=$REPO/adcensus_amd/lib/k8merge8/libadcensus_hip.so -- --steps 20
