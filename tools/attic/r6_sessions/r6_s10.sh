#!/bin/bash
# round 6, session 10: the corrected statistical tests, the 8-rank never-hang test, then the whole -m gpu suite
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06n; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_hip_rng.py -m gpu -x -q -k "adjacent or even_perm" > $out/pytest_rng.txt 2>&1; tail -5 $out/pytest_rng.txt
timeout 900 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "eight_ranks" -s > $out/pytest_8ranks.txt 2>&1; tail -8 $out/pytest_8ranks.txt
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; tail -8 $out/pytest_gpu.txt
