#!/bin/bash
# round 4, GPU call 10: K6 marching kernel, pipeline depth (2 / 3 steps in flight) x waves per workgroup (8 / 16): parity of every
# variant library, then the wta stage on the same box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
B="--no-cpu-baseline --no-extra-legs"
run() { TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4g_$TAG.json 2> $O/r4g_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4g_$TAG.json"))
    print("%-22s rc=$rc  %.1f pairs/s  %.3f ms  wta %.4f ms" % ("$TAG", d["value"], d["ms_per_step"], d["stage_ms"]["wta"]))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
for v in k6d3w8 k6d2w16 k6d3w16; do
  L=ADC_HIP_LIB=$REPO/adcensus_amd/lib/$v/libadcensus_hip.so
  env $L timeout 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_api.py -m gpu -x -q -k "stage_parity or right_wta" 2>&1 | tail -3 > $O/r4_gpu_pytest_$v.log; echo "$v: $(tail -1 $O/r4_gpu_pytest_$v.log)"
done
for rep in 1 2; do
  run noise_d2w8_$rep X=1 -- --steps 20
  for v in k6d3w8 k6d2w16 k6d3w16; do
    run noise_${v}_$rep ADC_HIP_LIB=$REPO/adcensus_amd/lib/$v/libadcensus_hip.so -- --steps 20
  done
done
run kitti_d2w8 X=1 -- --width 1242 --height 375 --steps 30
for v in k6d3w8 k6d2w16 k6d3w16; do
  run kitti_$v ADC_HIP_LIB=$REPO/adcensus_amd/lib/$v/libadcensus_hip.so -- --width 1242 --height 375 --steps 30
done
