#!/bin/bash
# round 6, session 9: the whole -m gpu suite, then the round's evidence set
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06m; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $out/pytest_gpu.txt 2>&1; tail -14 $out/pytest_gpu.txt
bash tools/r6_final_profiles.sh r06m > $out/final.log 2>&1; tail -30 $out/final.log
