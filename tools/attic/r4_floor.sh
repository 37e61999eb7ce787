#!/bin/bash
# launch floor of both launches at config 2: cut builds that return at once (S9 / F9), per-launch events and whole-call rate through the AQL queue
export PYTHONPATH=$GRAFT_REPO_ROOT HENS_DEBUG_NOFLIP=1
for L in d32 d32s9 d32f9 d32s1; do
  export HENS_LIB=$GRAFT_REPO_ROOT/build_ab/libhens_$L.so
  echo -n "$L events: "; timeout 200 python tools/quick_bench.py --steps 2000 --prof 1 2>&1 | sed -n 2,2p | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read()); print('first %.2f us  second %.2f us' % (d['stretch_ms']/d['n_stretch']*1e3, d['fused_ms']/d['n_fused']*1e3))"
  echo -n "$L aql: "; HENS_STEP_EVENTS= timeout 200 python tools/short_call.py 2000 2>&1 | tail -1
done
