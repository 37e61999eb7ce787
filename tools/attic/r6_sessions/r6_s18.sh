#!/bin/bash
# round 6, session 18: k_stretch2<PIPE> tests (forced onto small grids)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06p; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 2400 python -m pytest tests/test_hip_pipeline.py -q -k "persistent_pipelined" > $out/pytest_pipe2.txt 2>&1; tail -8 $out/pytest_pipe2.txt
