#!/bin/bash
# config 4 with the shipped library against ab_live/libhens_base.so, three alternations
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2 3; do for l in base new; do
  if [ $l = base ]; then export HENS_LIB=$R/ab_live/libhens_base.so; else unset HENS_LIB; fi
  echo -n "$l "; python tools/probe/rj_rung_leaves.py 2>&1 | grep us/iter | cut -c1-16 | tr '\n' ' '; echo
done; done
