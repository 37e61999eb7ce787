#!/bin/bash
# round 6, session 30: spread of the lazy-State timing test (five runs) and of the 8-rank W = 2048 test (three runs)
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for i in 1 2 3 4 5; do python -m pytest tests/test_hip_sampler.py -q -s -k "under_150" 2>&1 | grep -o "numpy loop.*config 2\|passed\|failed" | tr '\n' ' '; echo; done
for i in 1 2 3; do python -m pytest tests/test_hip_fullsize.py -q -k "eight_ranks" 2>&1 | tail -1; done
