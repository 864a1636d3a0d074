#!/bin/bash
# round 4, GPU call 12: what is left of a full-ring aggregation pass when its memory streams cost nothing?  Libraries built with
# -DRR2_FAKE_MEM=1 (steady-state loads and stores of a wave go to one address), 2 (loads only), 3 (stores only); garbage results.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
: > $O/r4_k4_fake_mem_raw.txt
for wl in structured noise; do
  for v in real k4fake1 k4fake2 k4fake3; do
    L=X=1; [ $v != real ] && L=ADC_HIP_LIB=$REPO/adcensus_amd/lib/$v/libadcensus_hip.so
    echo "== $wl $v" | tee -a $O/r4_k4_fake_mem_raw.txt
    env $L ADC_AGG_SMALL_L=0 timeout 120 python tools/gpu_k4_fake_mem.py $wl 2>&1 | tail -3 | tee -a $O/r4_k4_fake_mem_raw.txt
  done
done
