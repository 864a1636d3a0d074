#!/bin/bash
# SQ counters of the region-voting chain's kernel on the structured pair (separate --pmc passes, kernel-trace only);
# summary per duration class (heavy first rounds / tail rounds): python tools/pmc_irv_summary.py
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rm -rf "$REPO/gpurun_out/pmcirv_$i"
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmcirv_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs --workload structured > "$REPO/gpurun_out/pmcirv_$i.log" 2>&1; echo "pass $i rc=$?"
done
cd "$REPO"; python tools/pmc_irv_summary.py
