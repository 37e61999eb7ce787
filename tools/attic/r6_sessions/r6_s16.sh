#!/bin/bash
# round 6, session 16: k_stretch2<PIPE> - parity of the ranks, then the rank's launches beside the lone ladder
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06p; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 1500 python -m pytest tests/test_hip_pipeline.py -q -x -k "persistent_pipelined" > $out/pytest_pipe2.txt 2>&1; tail -5 $out/pytest_pipe2.txt
for d in 0 1; do PIPE_DELAY=$d python tools/pipe_prof.py 8 16384 64 200; done 2>&1 | grep -v amdgpu.ids
HENS_NO_TILE2_PIPE=1 python tools/pipe_prof.py 8 16384 64 200 2>&1 | grep -v amdgpu.ids
python tools/pipe_prof.py 8 16384 64 200 2>&1 | grep -v amdgpu.ids
HENS_NO_TILE2_PIPE=1 python tools/pipe_prof.py 8 16384 64 200 2>&1 | grep -v amdgpu.ids
