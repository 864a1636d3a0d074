#!/bin/bash
# round 5: the last aggregation pass inside the L->R scanline pass (k_scanline_seg_agg): correctness + same-box A/B
O=gpurun_out/r5_20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "digests" > $O/pytest_digests.log 2>&1; echo "digests rc=$?"; tail -3 $O/pytest_digests.log
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_random.py -x -q > $O/pytest_api.log 2>&1; echo "api rc=$?"; tail -3 $O/pytest_api.log
B="--no-cpu-baseline --no-extra-legs --steps 20"
for rep in 1 2; do
for V in "ADC_FUSE_AGG_SO=0" "ADC_FUSE_AGG_SO=1"; do
  env $V timeout 300 python bench.py $B --workload noise > $O/b.json 2>/dev/null
  python - "$V" <<'P' | tee -a $O/ab_agg_so_fusion.txt
import json, sys
o = json.load(open('gpurun_out/r5_20/b.json'))
print(sys.argv[1], "pairs/s %.1f" % o['value'], "aggregate %.3f scanline %.3f ms" % (o['stage_ms']['aggregate'], o['stage_ms']['scanline']), "K4 launch %.4f" % o['roofline']['avg_launch_ms'], "ok" if o['farm_check']['ok'] else "MISMATCH", o['farm_check'].get('reference_mismatches'))
P
done
done
