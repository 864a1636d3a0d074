#!/bin/bash
# round 4, GPU call 4: K8 TAIL mode -- correctness (tail-mode variants, stage tests, one full-size structured pair with the mode on),
# then a sweep of threshold / schedule on the structured 1080p and KITTI-size benches (same box)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 500 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "tail_mode or voting_chain or async_pipeline" 2>&1 | tail -8 > $O/r4_gpu_pytest_4.log; cat $O/r4_gpu_pytest_4.log
grep -q " passed" $O/r4_gpu_pytest_4.log && ! grep -q "failed\|error" $O/r4_gpu_pytest_4.log || { echo "TESTS NOT GREEN -- stopping"; exit 1; }
ADC_IRV_TAIL=1024 timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_stages.py -m gpu -x -q -k "kitti or middlebury" 2>&1 | tail -6 > $O/r4_gpu_pytest_4b.log; cat $O/r4_gpu_pytest_4b.log
grep -q " passed" $O/r4_gpu_pytest_4b.log && ! grep -q "failed\|error" $O/r4_gpu_pytest_4b.log || { echo "TESTS (tail on) NOT GREEN -- stopping"; exit 1; }
B="--no-cpu-baseline --no-extra-legs"
run() { # tag, env..., -- bench args
  TAG=$1; shift; ENVV=(); while [ "$1" != "--" ]; do ENVV+=("$1"); shift; done; shift
  env "${ENVV[@]}" timeout 120 python bench.py $B "$@" > $O/r4d_$TAG.json 2> $O/r4d_$TAG.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/r4d_$TAG.json"))
    fb = d.get("async_fallbacks", {})
    print("%-30s rc=$rc  %.1f pairs/s  %.3f ms  refine %.3f  voting budget %s continuations %s" % ("$TAG", d["value"], d["ms_per_step"], d["stage_ms"]["refine"], fb.get("voting_chain_budget"), fb.get("voting_continuations")))
except Exception as e:
    print("$TAG rc=$rc unreadable:", e)
PY
}
S="--workload structured --steps 10"
run s_off_1 ADC_IRV_TAIL=0 -- $S
for TM in 64 256 1024 4096; do run s_tail${TM} ADC_IRV_TAIL=$TM -- $S; done
run s_off_2 ADC_IRV_TAIL=0 -- $S
for T1 in 600 750 1100 1400; do run s_tail1024_t1_$T1 ADC_IRV_TAIL=1024 ADC_IRV_TAIL_T1=$T1 -- $S; done
for TS in 200 300 600 800; do run s_tail1024_ts_$TS ADC_IRV_TAIL=1024 ADC_IRV_TAIL_TS=$TS -- $S; done
for R in 4 8 24; do run s_tail1024_r$R ADC_IRV_TAIL=1024 ADC_IRV_TAIL_ROUNDS=$R -- $S; done
run s_tail4096_r24_ts300 ADC_IRV_TAIL=4096 ADC_IRV_TAIL_ROUNDS=24 ADC_IRV_TAIL_TS=300 -- $S
run s_off_3 ADC_IRV_TAIL=0 -- $S
K="--width 1242 --height 375 --workload structured --steps 30"
run k_off ADC_IRV_TAIL=0 -- $K
for TM in 64 256 1024; do run k_tail$TM ADC_IRV_TAIL=$TM -- $K; done
run k_tail256_ts300 ADC_IRV_TAIL=256 ADC_IRV_TAIL_TS=300 ADC_IRV_TAIL_T1=700 -- $K
run n_off ADC_IRV_TAIL=0 -- --steps 20
run n_tail1024 ADC_IRV_TAIL=1024 -- --steps 20
cd /tmp && export TMPDIR=/tmp
TAG=structured_tail1024
rm -rf "$REPO/$O/prof_$TAG"
ADC_IRV_TAIL=1024 timeout 120 rocprofv3 --kernel-trace --stats -d "$REPO/$O/prof_$TAG" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 $B --workload structured > "$REPO/$O/rocprof_$TAG.log" 2>&1; echo "rocprof $TAG rc=$?"
cd "$REPO"
DB=$(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | tail -1)
timeout 120 python tools/irv_trace_summary.py $DB > $O/r4d_irv_chain_structured_tail1024.txt 2>&1; head -8 $O/r4d_irv_chain_structured_tail1024.txt
