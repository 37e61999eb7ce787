#!/bin/bash
# Ceiling of "launch 2's packet without the barrier bit"  - TIMING ONLY, the chain is wrong: dev build, HENS_AQL_NOBAR=1
# (second launch), 2 (first launch of the next iteration), 3 (both); whole-iteration rate of long calls on the AQL queue
export PYTHONPATH=$GRAFT_REPO_ROOT HENS_LIB=$GRAFT_REPO_ROOT/ab_live/libhens_${1:-nb32}.so HENS_DEBUG_NOFLIP=1
for rep in 1 2 3; do for nb in 0 1 2 3; do
  echo -n "HENS_AQL_NOBAR=$nb: "; HENS_AQL_NOBAR=$nb timeout 200 python tools/short_call.py 2000 2>&1 | tail -1
done; done
