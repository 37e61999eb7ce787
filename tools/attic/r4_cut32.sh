#!/bin/bash
# cumulative phase times of both launches at config 2 by cut builds (results wrong, HENS_DEBUG_NOFLIP keeps the state addressable)
export PYTHONPATH=$GRAFT_REPO_ROOT HENS_DEBUG_NOFLIP=1
for L in d32 d32s1 d32s2 d32s3 d32s4 d32f1 d32f2 d32f3 d32f4 d32f5; do
  export HENS_LIB=$GRAFT_REPO_ROOT/build_ab/libhens_$L.so
  echo -n "$L: "; timeout 200 python tools/quick_bench.py --steps 2000 --prof 1 2>&1 | sed -n 2,2p | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read()); print('first %.2f us  second %.2f us' % (d['stretch_ms']/d['n_stretch']*1e3, d['fused_ms']/d['n_fused']*1e3))"
done
