#!/bin/bash
# round 6, session 15: the round's final evidence set on the final library (tag r06z) + the whole -m gpu suite
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06z; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; tail -6 $out/pytest_gpu.txt
bash tools/r6_final_profiles.sh r06z > $out/final.log 2>&1; tail -25 $out/final.log
