#!/bin/bash
# cumulative phase times of both launches at 8 x 16384 x 64 by cut builds (results wrong, HENS_DEBUG_NOFLIP keeps the state addressable)
export PYTHONPATH=$GRAFT_REPO_ROOT HENS_DEBUG_NOFLIP=1
for L in d64 d64s1 d64s2 d64s3 d64s4 d64f1 d64f2 d64f3 d64f4; do
  export HENS_LIB=$GRAFT_REPO_ROOT/build_ab/libhens_$L.so
  echo -n "$L: "; timeout 200 python tools/quick_bench.py --T 8 --W 16384 --D 64 --steps 400 --prof 1 2>&1 | sed -n 2,2p | cut -c1-400
done
