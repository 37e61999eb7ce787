#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
export HENS_LIB=$PWD/build_ab/libhens_rj.so
python tools/trace_rj.py 4 > $O/trace_mh.txt 2>&1
python tools/trace_rj.py 5 > $O/trace_bd.txt 2>&1
tail -n 3 $O/trace_mh.txt $O/trace_bd.txt
