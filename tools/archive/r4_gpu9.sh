#!/bin/bash
# round 4, GPU call 9: K6 right-view WTA marching along rows (k_wta_right_march) against the band kernel (ADC_WTA_MARCH=0):
# parity (stage cases, forced launch plans), then same-box A/B of the wta stage and the kernel durations
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_api.py -m gpu -x -q -k "stage_parity or right_wta or middlebury" 2>&1 | tail -40 > $O/r4_gpu_pytest_k6.log; cat $O/r4_gpu_pytest_k6.log
grep -q " passed" $O/r4_gpu_pytest_k6.log && ! grep -q "failed\|error" $O/r4_gpu_pytest_k6.log || { echo "TESTS NOT GREEN -- stopping"; exit 1; }
B="--no-cpu-baseline --no-extra-legs"
run() { TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4f_$TAG.json 2> $O/r4f_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4f_$TAG.json"))
    print("%-22s rc=$rc  %.1f pairs/s  %.3f ms  wta %.4f ms" % ("$TAG", d["value"], d["ms_per_step"], d["stage_ms"]["wta"]))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
for rep in 1 2; do
  run noise_band_$rep ADC_WTA_MARCH=0 -- --steps 20
  run noise_march_$rep X=1 -- --steps 20
done
run struct_band ADC_WTA_MARCH=0 -- --workload structured --steps 10
run struct_march X=1 -- --workload structured --steps 10
run kitti_band ADC_WTA_MARCH=0 -- --width 1242 --height 375 --steps 30
run kitti_march X=1 -- --width 1242 --height 375 --steps 30
run kitti_march_nseg1 ADC_WTA_NSEG=1 -- --width 1242 --height 375 --steps 30
run noise_march_nseg1 ADC_WTA_NSEG=1 -- --steps 20
# kernel durations
export TMPDIR=/tmp
for v in 0 1; do
  ADC_WTA_MARCH=$v timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_k6_$v -o k6 -- python bench.py $B --steps 5 --warmup 2 > /dev/null 2>&1
  f=$(find $O/prof_k6_$v -name "*kernel_stats.csv" | head -1)
  echo "ADC_WTA_MARCH=$v:"; grep -i "wta" "$f" | cut -c1-200
done
