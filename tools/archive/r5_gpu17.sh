#!/bin/bash
# round 5: structured pairs in flight (2 / 3 / 4 / 6) and the shared bandwidth lane, new K8
O=gpurun_out/r5_17; mkdir -p $O
B="--no-cpu-baseline --no-extra-legs --steps 24 --workload structured"
for rep in 1 2; do
for V in "X=0 --inflight 1" "X=0 --inflight 2" "X=0 --inflight 3" "X=0 --inflight 4" "X=0 --inflight 6" "ADC_SHARED_HEAVY=1 --inflight 3" "ADC_SHARED_HEAVY=1 --inflight 4"; do
    E=${V%% *}; A=${V#* }
    env $E timeout 300 python bench.py $B $A > $O/b.json 2>/dev/null
    python - "$V" <<'P' | tee -a $O/structured_inflight.txt
import json, sys
o = json.load(open('gpurun_out/r5_17/b.json'))
print(sys.argv[1], "pairs/s %.1f" % o['value'], "ok" if o['farm_check']['ok'] else "MISMATCH", o['async_fallbacks'].get('voting_continuations'))
P
done
done
