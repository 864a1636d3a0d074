#!/bin/bash
# round 5: timing proxy of the K4(H4) -> K5(L->R) fusion: the scanline passes with the aggregation arithmetic of a short-arm ring
# added to every step (variant build -DSO_PROXY_AGG=1; results unchanged), against the product, same box, interleaved
O=gpurun_out/r5_19; mkdir -p $O
B="--no-cpu-baseline --no-extra-legs --steps 20"
for rep in 1 2 3; do
for V in base so_proxy; do
  L=adcensus_amd/lib/$V/libadcensus_hip.so; [ $V = base ] && L=adcensus_amd/lib/libadcensus_hip.so
  ADC_HIP_LIB=$L timeout 300 python bench.py $B --workload noise > $O/b.json 2>/dev/null
  python - "$V" <<'P' | tee -a $O/ab_fusion_proxy.txt
import json, sys
o = json.load(open('gpurun_out/r5_19/b.json'))
print(sys.argv[1], "pairs/s %.1f" % o['value'], "aggregate %.3f scanline %.3f ms" % (o['stage_ms']['aggregate'], o['stage_ms']['scanline']), "K4 launch %.4f" % o['roofline']['avg_launch_ms'], "ok" if o['farm_check']['ok'] else "MISMATCH")
P
done
done
