#!/bin/bash
# round 3, first diagnostic: baseline, kernarg placement knob, start/end skew of a launch's workgroups
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_diag1; mkdir -p $O
for i in 1 2; do
  echo -n "default: "; timeout 120 python tools/quick_bench.py --steps 4000 2>&1 | grep -o "[0-9.]* us/iter"
  echo -n "DEV_KERNARG=1: "; HIP_FORCE_DEV_KERNARG=1 timeout 120 python tools/quick_bench.py --steps 4000 2>&1 | grep -o "[0-9.]* us/iter"
  echo -n "DEV_KERNARG=0: "; HIP_FORCE_DEV_KERNARG=0 timeout 120 python tools/quick_bench.py --steps 4000 2>&1 | grep -o "[0-9.]* us/iter"
done
echo "--- per-launch events"; timeout 120 python tools/quick_bench.py --prof 1 2>&1 | sed -n 1,3p
echo "--- skew"; HENS_LIB=$GRAFT_REPO_ROOT/build_ab/libhens_rt.so timeout 200 python tools/trace_skew.py > $O/skew.txt 2>&1; cat $O/skew.txt
echo "--- bench"; timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
