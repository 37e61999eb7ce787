#!/bin/bash
O=gpurun_out/r4c; mkdir -p $O
build_ab/step_floor 0 > $O/step_floor.txt 2>&1
build_ab/step_floor 1 > $O/step_floor_devsync.txt 2>&1
cat $O/step_floor.txt $O/step_floor_devsync.txt
