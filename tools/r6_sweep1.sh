#!/bin/bash
# round 6, call 2: the new GPU tests (fault injection, the reference's own caller, the probed sweep layout), the default bench line with
# its new legs, and two same-box sweeps over existing switches: aggregation chunking at the KITTI size (review item 3) and the
# number of verified scanline row segments at 1080p (item 6).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_faults.py tests/test_reference_caller.py tests/test_gpu_api.py tests/test_gpu_stages.py tests/test_gpu_fused_tail.py -m gpu -x -q --durations=5 2>&1 | tail -14 > $O/r6b_gpu_pytest.log; cat $O/r6b_gpu_pytest.log
timeout 500 python bench.py > $O/r6b_bench_default.json 2> $O/r6b_bench_default.err; echo "default rc=$?"; python tools/bench_brief.py $O/r6b_bench_default.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6b_bench_default.json").read())
print("match_host", json.dumps(d.get("match_host"))[:900])
print("cpu_all", json.dumps(d.get("cpu_baseline_all_cores"))[:500])
print("structured tp/host", d["structured"].get("throughput_mode", {}).get("value"), d["structured"].get("host_inclusive", {}).get("value"), d["structured"].get("voting"))
PY
B="--no-cpu-baseline --no-extra-legs"
K="--width 1242 --height 375 --steps 40 $B"
run() { tag=$1; shift; env "$@" timeout 120 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
for WL in structured noise; do
  ARGS="$K --workload $WL"
  run kitti_${WL}_default X=1
  run kitti_${WL}_h1242 ADC_AGG_HCHUNK=1242
  run kitti_${WL}_h621 ADC_AGG_HCHUNK=621
  run kitti_${WL}_h414 ADC_AGG_HCHUNK=414
  run kitti_${WL}_v375 ADC_AGG_VCHUNK=375
  run kitti_${WL}_v375_h621 ADC_AGG_VCHUNK=375 ADC_AGG_HCHUNK=621
  run kitti_${WL}_v375_h414 ADC_AGG_VCHUNK=375 ADC_AGG_HCHUNK=414
  run kitti_${WL}_rr1_default ADC_AGG_RR2=0
  run kitti_${WL}_rr1_seg1 ADC_AGG_RR2=0 ADC_AGG_HSEG=1 ADC_AGG_VSEG=1
  run kitti_${WL}_rr1_h2v1 ADC_AGG_RR2=0 ADC_AGG_HSEG=2 ADC_AGG_VSEG=1
  run kitti_${WL}_rr1_h3v1 ADC_AGG_RR2=0 ADC_AGG_HSEG=3 ADC_AGG_VSEG=1
  run kitti_${WL}_default_again X=1
done
for WL in noise structured; do
  ARGS="--steps 20 $B --workload $WL"
  run so_${WL}_default X=1
  run so_${WL}_seg3 ADC_SO_SEG=3
  run so_${WL}_seg4 ADC_SO_SEG=4
  run so_${WL}_seg2_wpb2 ADC_SO_WPB_ROW=2
  run so_${WL}_seg4_wpb2 ADC_SO_SEG=4 ADC_SO_WPB_ROW=2
  run so_${WL}_default_again X=1
done
