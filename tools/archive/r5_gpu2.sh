#!/bin/bash
# round 5, call 2: Infinity-Cache residency micro-benchmark, the C++ multi-GPU farm example on the one GPU of the box
O=gpurun_out/r5_2; mkdir -p $O
timeout 300 tools/ubench/l3_resident > $O/ubench_l3_resident.txt 2>&1; echo "l3 rc=$?"; cat $O/ubench_l3_resident.txt
timeout 300 adcensus_amd/bin/adcensus_farm_multi 24 > $O/farm_multi_1gpu.json 2> $O/farm_multi.err; echo "farm_multi rc=$?"; cat $O/farm_multi_1gpu.json; tail -3 $O/farm_multi.err
