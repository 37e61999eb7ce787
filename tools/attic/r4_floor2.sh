#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
for L in main d32; do
  if [ $L = main ]; then unset HENS_LIB; else export HENS_LIB=$GRAFT_REPO_ROOT/build_ab/libhens_$L.so; fi
  echo -n "$L aql: "; timeout 200 python tools/short_call.py 2000 2>&1 | tail -1
  echo -n "$L aql K=20: "; timeout 200 python tools/short_call.py 20 2>&1 | tail -1
done; done
