#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
for L in d64 d64mf; do echo "== $L"; HENS_LIB=$GRAFT_REPO_ROOT/build_ab/libhens_$L.so python tools/like_check.py 8 4096 64 2>&1 | tail -3; HENS_LIB=$GRAFT_REPO_ROOT/build_ab/libhens_$L.so python tools/like_check.py 4 1000 64 nonsym 2>&1 | tail -3; done
