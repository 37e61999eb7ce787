#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
export HENS_LIB=$PWD/build_ab/libhens_aql.so
timeout 300 python tools/aql_check.py > $O/aql_check.txt 2>&1; echo "rc=$?" >> $O/aql_check.txt
timeout 120 python tools/short_call.py > $O/short_aql.log 2>&1
HENS_NO_AQL=1 timeout 120 python tools/short_call.py > $O/short_hip.log 2>&1
for r in 1 2 4 8 16 64; do echo "ring every $r"; HENS_AQL_RING=$r timeout 120 python tools/short_call.py 2>&1 | grep -v amdgpu.ids; done > $O/short_ring.log 2>&1
cat $O/aql_check.txt $O/short_aql.log $O/short_hip.log $O/short_ring.log
