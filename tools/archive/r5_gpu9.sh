#!/bin/bash
# round 5, call 9: non-temporal volume accesses in K4 (A/B on one box, both 1080p workloads)
O=gpurun_out/r5_9; mkdir -p $O
B="--no-cpu-baseline --no-extra-legs --steps 20"
for rep in 1 2; do
for V in base vol_nt vol_nt_k45 vol_nt_all; do
  L=adcensus_amd/lib/$V/libadcensus_hip.so; [ $V = base ] && L=adcensus_amd/lib/libadcensus_hip.so
  for WL in noise structured; do
    ADC_HIP_LIB=$L timeout 300 python bench.py $B --workload $WL > $O/b.json 2>/dev/null
    python - "$V" "$WL" <<'P' | tee -a $O/ab_k4_nontemporal.txt
import json, sys
o = json.load(open('gpurun_out/r5_9/b.json'))
print(sys.argv[1], sys.argv[2], "pairs/s %.1f" % o['value'], "agg stage %.3f ms" % o['stage_ms']['aggregate'], "scanline %.3f wta %.3f" % (o['stage_ms']['scanline'], o['stage_ms']['wta']), "K4 launch %.4f ms frac %.3f" % (o['roofline']['avg_launch_ms'], o['roofline']['frac']), "ok" if o['farm_check']['ok'] else "MISMATCH")
P
  done
done
done
