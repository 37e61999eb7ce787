#!/bin/bash
O=gpurun_out/r4f; mkdir -p $O
export HENS_AQL_STATS=1
for v in "HENS_AQL_KA_POOL=fine" "HENS_AQL_KA_POOL=coarse" "HENS_AQL_KA_POOL=coarse HENS_AQL_ACQ_AGENT=1 HENS_AQL_FLUSH=1"; do
  echo "== $v"; env $v timeout 120 build_ab/step_floor_dev 1 2>&1 | grep -v "K =    [25]:\|K =   10\|K =   40"
done > $O/floor_variants2.txt 2>&1
export HENS_LIB=$PWD/build_ab/libhens_aql.so
for v in "HENS_AQL_KA_POOL=fine" "HENS_AQL_KA_POOL=coarse"; do echo "== $v"; env $v timeout 120 python tools/short_call.py 2>&1 | grep -v amdgpu.ids; done > $O/short_aql2.log 2>&1
cat $O/floor_variants2.txt $O/short_aql2.log
