#!/bin/bash
# Round 5, session 7: head __syncthreads() removed at D != 32 (slack table again: the step at inject = 1), phase-A cuts at config 2,
# pipeline tests
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5_s7; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_pipeline.py -x -q -m gpu 2>&1 | tail -2 > $out/pytest.txt; cat $out/pytest.txt
bash tools/pipe_slack.sh 8 16384 64 inj64 > $out/slack64.txt 2>&1; cut -c1-130 $out/slack64.txt
unset HENS_LIB
for rep in 1 2; do for d in 0 1; do PIPE_DELAY=$d timeout 200 python tools/pipe_prof.py 8 16384 64 200 2>&1 | grep -E "^pipe|^single"; done; done | cut -c1-150 > $out/pipe_rank.txt; cat $out/pipe_rank.txt
HENS_PIPE_STATS=1 timeout 300 python tools/time_pipeline.py local 2 16 16384 64 200 2>&1 | grep -v amdgpu | cut -c1-200 > $out/local2.txt; cat $out/local2.txt
bash tools/cut_phase_a.sh > $out/cut_phase_a.txt 2>&1; cat $out/cut_phase_a.txt
