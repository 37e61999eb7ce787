#!/bin/bash
# round 6, session 3: the whole -m gpu suite on the current tree
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06c; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > $out/pytest_gpu.txt 2>&1; tail -30 $out/pytest_gpu.txt
