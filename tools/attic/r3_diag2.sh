#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_diag2; mkdir -p $O
echo "--- launch tax"; timeout 120 build_ab/launch_tax 2>&1 | tee $O/launch_tax.txt
echo "--- k_iter at config 2 (HENS_ITER_MAX=2)"
for i in 1 2; do
  echo -n "two-launch: "; timeout 120 python tools/quick_bench.py --steps 4000 2>&1 | grep -o "[0-9.]* us/iter"
  echo -n "k_iter:     "; HENS_ITER_MAX=2 timeout 120 python tools/quick_bench.py --steps 4000 2>&1 | grep -o "[0-9.]* us/iter"
done
echo "--- k_iter skew"; HENS_ITER_MAX=2 HENS_LIB=$GRAFT_REPO_ROOT/build_ab/libhens_rt.so timeout 200 python tools/trace_skew.py > $O/skew_iter.txt 2>&1; tail -22 $O/skew_iter.txt
echo "--- no-plan bound (HENS_PLAN reuse?)"
