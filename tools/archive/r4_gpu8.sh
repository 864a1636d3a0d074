#!/bin/bash
# round 4, GPU call 8: K4 full ring with 16 entries in flight per wave (variant library adcensus_amd/lib/k4pf16, -DRR2_PF=16) against 8:
# stage parity of the variant, then same-box A/B on the structured pair and on the noise pair forced onto the full ring
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
V=ADC_HIP_LIB=$REPO/adcensus_amd/lib/k4pf16/libadcensus_hip.so
env $V timeout 480 python -m pytest tests/test_gpu_stages.py tests/test_gpu_fullsize.py -m gpu -x -q -k "stage_parity or kitti or full" 2>&1 | tail -5 > $O/r4_gpu_pytest_k4pf16.log; cat $O/r4_gpu_pytest_k4pf16.log
grep -q " passed" $O/r4_gpu_pytest_k4pf16.log && ! grep -q "failed\|error" $O/r4_gpu_pytest_k4pf16.log || { echo "TESTS NOT GREEN -- stopping"; exit 1; }
B="--no-cpu-baseline --no-extra-legs"
run() { TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4e_$TAG.json 2> $O/r4e_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4e_$TAG.json"))
    print("%-22s rc=$rc  %.1f pairs/s  %.3f ms  aggregate %.3f  K4 launch %.4f ms frac %.3f" % ("$TAG", d["value"], d["ms_per_step"], d["stage_ms"]["aggregate"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
for rep in 1 2 3; do
  run struct_pf8_$rep X=1 -- --workload structured --steps 10
  run struct_pf16_$rep $V -- --workload structured --steps 10
done
for rep in 1 2; do
  run noisefull_pf8_$rep ADC_AGG_SMALL_L=0 -- --steps 20
  run noisefull_pf16_$rep ADC_AGG_SMALL_L=0 $V -- --steps 20
done
run kitti_struct_pf8 X=1 -- --width 1242 --height 375 --workload structured --steps 30
run kitti_struct_pf16 $V -- --width 1242 --height 375 --workload structured --steps 30
