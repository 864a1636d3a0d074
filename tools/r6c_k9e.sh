#!/bin/bash
# round 6, second session: K9 trip escalation (first / second / following trips) and v_sad_u8 colour distance -- same box, interleaved
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
O=gpurun_out; L=$GRAFT_REPO_ROOT/adcensus_amd/lib
B="--no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $ARGS > $O/sw_$tag.json 2> $O/sw_$tag.err; python tools/bench_brief.py $O/sw_$tag.json; }
for rep in 1 2; do
  ARGS="--steps 20 $B --workload noise"
  run k9e_noise_prev_$rep ADC_HIP_LIB=$L/r6prev/libadcensus_hip.so
  for v in k9e_244x k9e_248 k9e_148 k9e_246 k9e_228; do run k9e_noise_${v}_$rep ADC_HIP_LIB=$L/$v/libadcensus_hip.so; done
done
ARGS="--width 1242 --height 375 --steps 40 $B --workload noise"
for v in k9e_244x k9e_248 k9e_246; do run k9e_kitti_noise_${v} ADC_HIP_LIB=$L/$v/libadcensus_hip.so; done
timeout 900 python -m pytest tests/test_gpu_stages.py -m gpu -x -q 2>&1 | tail -2
# (profiling level 2 in bench.py: the default library, full default line without the CPU baseline)
timeout 600 python bench.py --no-cpu-baseline > $O/k9e_bench_default.json 2> $O/k9e_bench_default.err; python tools/bench_brief.py $O/k9e_bench_default.json
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/k9e_bench_default.json") if l.startswith("{")][-1])
print("stage", d["stage_ms"], "latency", d["ms_per_pair_latency"], "ms/step", d["ms_per_step"])
print("structured", d.get("structured", {}).get("value"), d.get("structured", {}).get("stage_ms"))
print("match_host", json.dumps(d.get("match_host"))[:600])
PY
