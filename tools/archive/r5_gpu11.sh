#!/bin/bash
O=gpurun_out/r5_11; mkdir -p $O
timeout 300 tools/ubench/row_pitch > $O/ubench_row_pitch.txt 2>&1; cat $O/ubench_row_pitch.txt
timeout 300 tools/ubench/row_pitch >> $O/ubench_row_pitch.txt 2>&1; tail -9 $O/ubench_row_pitch.txt
