#!/bin/bash
# DEV: config 4 timing of dev builds + the RJ GPU tests against one of them
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2; do for l in ${LIBS:-wt2 v2}; do echo $l; HENS_LIB=$R/ab_live/libhens_$l.so python tools/probe/rj_rung_leaves.py 2>&1 | grep us/iter | cut -c1-20; done; done
if [ -n "$TESTLIB" ]; then HENS_LIB=$R/ab_live/libhens_$TESTLIB.so timeout 900 python -m pytest tests/test_hip_rj.py -x -q -m gpu 2>&1 | tail -15 | cut -c1-220; fi
