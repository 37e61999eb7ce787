#!/bin/bash
# DEV A/B: config 4 iteration time per build, then the phase stamps of both launch kinds (stride-trace build)
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2; do for l in ${LIBS:-hd wt2}; do echo $l; HENS_LIB=$R/ab_live/libhens_$l.so python tools/probe/rj_rung_leaves.py 2>&1 | grep us/iter | cut -c1-20; done; done
if [ -n "$TRACE" ]; then for m in 4 5; do HENS_LIB=$R/ab_live/libhens_$TRACE.so python tools/trace_rj.py $m rungs 2>&1 | grep -v amdgpu | grep "mode\|rung [4567]"; done; fi
