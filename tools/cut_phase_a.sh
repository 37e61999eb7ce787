#!/bin/bash
# Where the first launch's time in front of its row gathers goes at config 2: cumulative launch-level times of cut
# builds on the AQL queue (whole-iteration rate of long calls; the second launch is unchanged; results are wrong, HENS_DEBUG_NOFLIP
# keeps the state addressable).  Build first:  for c in 9 10 11 12 13 1 2; do tools/devbuild.sh c$c -DHENS_CUT_S=$c; done; tools/devbuild.sh c0
export PYTHONPATH=$GRAFT_REPO_ROOT HENS_DEBUG_NOFLIP=1
for rep in 1 2 3; do for L in c0 c9 c10 c11 c12 c13 c1 c2; do
  export HENS_LIB=$GRAFT_REPO_ROOT/ab_live/libhens_$L.so
  echo -n "$L: "; timeout 200 python tools/short_call.py 2000 2>&1 | tail -1
done; done
