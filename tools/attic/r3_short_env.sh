#!/bin/bash
# Short-call (driver shape: blocks of 20 iterations between synchronisations) under host-side wait policies, and the
# one-launch iteration forced onto config 2.
out=gpurun_out/${1:-s3b}; mkdir -p $out
run() { echo "== $*" >> $out/short_env.log; env "$@" python tools/short_call.py 20 >> $out/short_env.log 2>&1; }
run X=1
run ROC_ACTIVE_WAIT_TIMEOUT=1000
run HSA_ENABLE_INTERRUPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=1000 HSA_ENABLE_INTERRUPT=0
run HIP_FORCE_DEV_KERNARG=1
run GPU_MAX_HW_QUEUES=1
echo "== k_iter forced on config 2" >> $out/short_env.log
HENS_ITER_MAX=2 python tools/quick_bench.py --prof 0 >> $out/short_env.log 2>&1
HENS_ITER_MAX=2 python tools/quick_bench.py --prof 1 >> $out/short_env.log 2>&1
python tools/quick_bench.py --prof 0 >> $out/short_env.log 2>&1
cat $out/short_env.log
