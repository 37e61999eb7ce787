#!/bin/bash
# cumulative phase times at config 2 on the AQL queue (no events): whole-iteration rate of cut builds, HENS_DEBUG_NOFLIP everywhere
export PYTHONPATH=$GRAFT_REPO_ROOT HENS_DEBUG_NOFLIP=1
for L in d32 d32s9 d32s1 d32s2 d32s3 d32s4 d32f9 d32f1 d32f2 d32f3 d32f4 d32f5; do
  export HENS_LIB=$GRAFT_REPO_ROOT/build_ab/libhens_$L.so
  echo -n "$L: "; timeout 200 python tools/short_call.py 2000 2>&1 | tail -1
done
