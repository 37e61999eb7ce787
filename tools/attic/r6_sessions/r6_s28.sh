#!/bin/bash
# round 6, session 28: k_stretch2<PIPE> not where a launch waits for other ranks' counts - the selection test, the forced parity tests
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
timeout 2400 python -m pytest tests/test_hip_pipeline.py -q -k "persistent_pipelined or keep_the_rounds" 2>&1 | tail -8
