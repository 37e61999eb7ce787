#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O
export HENS_AQL_STATS=1
timeout 120 build_ab/step_floor_dev 1 > $O/floor_aql.txt 2>&1
HENS_AQL_HOST_KERNARG=1 timeout 120 build_ab/step_floor_dev 1 > $O/floor_aql_hostka.txt 2>&1
export HENS_LIB=$PWD/build_ab/libhens_aql.so
timeout 120 python tools/short_call.py > $O/short_aql.log 2>&1
timeout 300 python tools/aql_check.py > $O/aql_check.txt 2>&1; echo "rc=$?" >> $O/aql_check.txt
cat $O/floor_aql.txt $O/floor_aql_hostka.txt $O/short_aql.log $O/aql_check.txt
