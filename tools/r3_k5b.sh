#!/bin/bash
# K5 interior chunks with pinned prefetch slots: full GPU test tier first (stops at the first failure, nothing else runs then),
# afterwards the scanline stage time with the short form off / on at both sizes (same box, alternating)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14 > $O/k5b_pytest.log; cat $O/k5b_pytest.log
grep -q " passed" $O/k5b_pytest.log && ! grep -q "failed\|error" $O/k5b_pytest.log || { echo "TESTS NOT GREEN -- stopping"; exit 1; }
for rep in 1 2; do
 for F in 0 1; do
  for CFG in "1920 1080 noise" "1242 375 noise" "1242 375 structured"; do
    set -- $CFG
    ADC_SO_FAST=$F timeout 150 python bench.py --width $1 --height $2 --workload $3 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/k5b_$1_$3_f${F}_$rep.json 2> $O/k5b_err.txt || { echo "bench failed"; tail -3 $O/k5b_err.txt; exit 1; }
    python - <<PY
import json
d=json.load(open("$O/k5b_$1_$3_f${F}_$rep.json"))
print("fast=$F", "$1x$2", "$3", "pairs/s", round(d["value"],1), "scanline ms", d["stage_ms"]["scanline"], "farm_check", d.get("farm_check"))
PY
  done
 done
done
