#!/bin/bash
# round 6, session 11: k_stretch2 (persistent, software-pipelined first launch) - bit identity and timing
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06p; mkdir -p $out; cd $R; export PYTHONPATH=$R
{ timeout 600 python tools/tile2_check.py 4 512 64 dense 200
  timeout 600 python tools/tile2_check.py 8 2048 64 diag 300
  timeout 600 python tools/tile2_check.py 8 2048 64 rosen 300
  timeout 600 python tools/tile2_check.py 8 16384 64 dense 300 2
  timeout 600 python tools/tile2_check.py 8 16384 64 diag 300
  timeout 600 python tools/tile2_check.py 16 16384 64 dense 100
  timeout 600 python tools/tile2_check.py 8 8192 64 dense 300; } 2>&1 | grep -v amdgpu.ids | tee $out/tile2_check.txt
