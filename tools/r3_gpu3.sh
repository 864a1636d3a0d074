#!/bin/bash
# round-3 GPU session 3: chunked pair-register ring: parity, A/B of the chunk length, short-span floor, SQ counters
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/g3_pytest.log; cat $O/g3_pytest.log
run() { # tag, env...
  local tag=$1; shift
  env "$@" timeout 600 python bench.py --workload ${WL:-structured} --steps 10 --no-cpu-baseline --no-extra-legs > $O/g3_$tag.json 2> $O/g3_$tag.err
  python - "$tag" <<'PY'
import json, sys
n = sys.argv[1]
try:
    o = json.loads(open("gpurun_out/g3_%s.json" % n).read().strip().splitlines()[-1])
    print(n, o["value"], o["stage_ms"], o["roofline"]["avg_launch_ms"], o["roofline"]["hbm_frac"], "farm", o["farm_check"]["ok"])
except Exception as e:
    print(n, "ERR", e)
PY
}
run struct_default X=1
run struct_hchunk384 ADC_AGG_HCHUNK=384
run struct_hchunk640 ADC_AGG_HCHUNK=640
run struct_vchunk1013 ADC_AGG_VCHUNK=1013
run struct_rr2off ADC_AGG_RR2=0
WL=noise run noise_fullring ADC_AGG_SMALL_L=0
WL=noise run noise_fullring_rr2off ADC_AGG_SMALL_L=0 ADC_AGG_RR2=0
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rm -rf "$REPO/$O/pmcst_$i"
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$REPO/$O/pmcst_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --workload structured > "$REPO/$O/pmcst_$i.log" 2>&1; echo "sq pass $i rc=$?"
done
cd "$REPO"; python tools/pmc_sq_summary.py $O/pmcst_ > $O/r3_sq_all_structured.md 2>&1; head -16 $O/r3_sq_all_structured.md
