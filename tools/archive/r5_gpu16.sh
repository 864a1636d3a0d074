#!/bin/bash
# round 5: K4 launch-shape knobs of the small-ring pair kernels (noise pair), interleaved with the default
O=gpurun_out/r5_16; mkdir -p $O
B="--no-cpu-baseline --no-extra-legs --steps 20"
for rep in 1 2; do
for V in "X=0" "ADC_AGG_ASSUME_MARGIN=0" "ADC_AGG_VSEG=2" "ADC_AGG_VSEG=3" "ADC_AGG_VSEG=6" "ADC_AGG_HSEG=3" "ADC_AGG_HSEG=6" "ADC_AGG_HSEG=8" "ADC_AGG_VSEG=8"; do
    env $V timeout 300 python bench.py $B --workload noise > $O/b.json 2>/dev/null
    python - "$V" <<'P' | tee -a $O/k4_shape_knobs.txt
import json, sys
o = json.load(open('gpurun_out/r5_16/b.json'))
print(sys.argv[1], "pairs/s %.1f" % o['value'], "agg stage %.3f ms" % o['stage_ms']['aggregate'], "K4 launch %.4f ms frac %.3f" % (o['roofline']['avg_launch_ms'], o['roofline']['frac']), "ok" if o['farm_check']['ok'] else "MISMATCH")
P
done
done
