#!/bin/bash
# round-3 GPU session 1: primitives of the two-disparities-per-lane register ring, copy ceilings of the K4 access shapes,
# H-pass fetch with one segment per row, SQ counters of every kernel of the noise Match
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
REPO="$GRAFT_REPO_ROOT"; O=gpurun_out
timeout 120 tools/ubench/idx_pk > $O/r3_ubench_idx_pk.txt 2>&1; cat $O/r3_ubench_idx_pk.txt
timeout 300 tools/ubench/stream_shapes > $O/r3_ubench_stream_shapes.txt 2>&1; cat $O/r3_ubench_stream_shapes.txt
cd /tmp && export TMPDIR=/tmp
for HS in 1 0; do
  rm -rf "$REPO/$O/pmc_hseg${HS}_FETCH_SIZE"
  ADC_AGG_HSEG=$HS timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$REPO/$O/pmc_hseg${HS}_FETCH_SIZE" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --workload structured > "$REPO/$O/pmc_hseg${HS}.log" 2>&1; echo "pmc hseg $HS rc=$?"
done
(cd "$REPO"; python - <<'PY'
import csv, glob, collections
for hs in (1, 0):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_hseg%d_FETCH_SIZE/**/pmc_counter_collection.csv" % hs, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and "k_agg_regring" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0]].append((float(r["Counter_Value"]) * 2048.0, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0))
    for k, v in sorted(agg.items()):
        print("HSEG=%d %-40s n=%d fetch(x2) %.4f GB  avg %.1f us" % (hs, k, len(v), sum(a for a, _ in v) / len(v) / 1e9, sum(b for _, b in v) / len(v)))
PY
) | tee "$REPO/$O/r3_hseg_fetch.txt"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rm -rf "$REPO/$O/pmcall_$i"
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$REPO/$O/pmcall_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > "$REPO/$O/pmcall_$i.log" 2>&1; echo "sq pass $i rc=$?"
done
cd "$REPO"; python tools/pmc_sq_summary.py $O/pmcall_ > $O/r3_sq_all_noise.md 2>&1; head -24 $O/r3_sq_all_noise.md
