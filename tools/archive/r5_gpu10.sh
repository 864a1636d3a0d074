#!/bin/bash
# round 5: K9 from LDS tiles (k_interpolate_tile): stage tests, A/B against the list kernel alone
O=gpurun_out/r5_10; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_stages.py tests/test_gpu_api.py tests/test_gpu_fullsize.py -x -q -k "not isolation" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
B="--no-cpu-baseline --no-extra-legs --steps 20"
for rep in 1 2; do
for V in "ADC_INTERP_TILE=0" "ADC_INTERP_TILE=1" "ADC_INTERP_TILE=1 ADC_INTERP_NS=2"; do
  for WL in noise structured; do
    env $V timeout 300 python bench.py $B --workload $WL > $O/b.json 2>/dev/null
    python - "$V" "$WL" <<'P' | tee -a $O/ab_k9_tile_lds.txt
import json, sys
o = json.load(open('gpurun_out/r5_10/b.json'))
print(sys.argv[1], sys.argv[2], "pairs/s %.1f" % o['value'], "refine %.3f ms" % o['stage_ms']['refine'], "K4 launch %.4f" % o['roofline']['avg_launch_ms'], "ok" if o['farm_check']['ok'] else "MISMATCH")
P
  done
done
done
