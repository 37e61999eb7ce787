#!/bin/bash
# three-way same-box A/B at config 2: libhens_base.so, the current library, libhens_short.so (4000 iterations, 4 rounds)
export PYTHONPATH=$GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/eryn_amd/lib
for i in 1 2 3 4; do
  for v in base cur short; do
    case $v in base) lib=$GRAFT_REPO_ROOT/ab_live/libhens_base.so;; cur) lib=$L/libhipensemble.so;; short) lib=$GRAFT_REPO_ROOT/ab_live/libhens_short.so;; esac
    echo -n "$v: "; HENS_LIB=$lib timeout 120 python tools/quick_bench.py --steps 4000 --prof 0 2>&1 | grep -o "[0-9.]* us/iter"
  done
done
