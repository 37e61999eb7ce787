#!/bin/bash
# round 6, session 6: reversible jump with a host-callable likelihood on the device path; RJ regression
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06f; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_rj_callable.py -m gpu -x -q > $out/pytest_callable.txt 2>&1; tail -25 $out/pytest_callable.txt
timeout 1200 python -m pytest tests/test_hip_rj.py tests/test_hip_sampler.py -m gpu -x -q > $out/pytest_rj.txt 2>&1; tail -5 $out/pytest_rj.txt
