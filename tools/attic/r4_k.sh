#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O
for L in aql mf mf_noload; do
export HENS_LIB=$PWD/build_ab/libhens_$L.so
echo "== $L"
timeout 120 python tools/short_call.py 2>&1 | grep "long call\|eng.sync"
timeout 120 python tools/trace_fused.py 16 4096 32 3 1 2>&1 | grep "phase durations\|lifetime mean"
timeout 120 python tools/trace_fused.py 16 4096 32 1 1 2>&1 | grep "phase durations\|lifetime mean"
done > $O/ab.txt 2>&1
cat $O/ab.txt
