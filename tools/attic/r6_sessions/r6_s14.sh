#!/bin/bash
# round 6, session 14: k_split1_pt2 (persistent, software-pipelined second launch) - bit identity and timing, per launch
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06s; mkdir -p $out; cd $R; export PYTHONPATH=$R
{ for w in 3 1 2; do
    export HENS_TILE2_LAUNCH=$w HENS_KEEP_ENV=1
    echo "== HENS_TILE2_LAUNCH=$w"
    timeout 600 python tools/tile2_check.py 8 512 64 dense 200
    timeout 600 python tools/tile2_check.py 8 2048 64 diag 300
    timeout 600 python tools/tile2_check.py 8 2048 64 rosen 300
    timeout 600 python tools/tile2_check.py 8 16384 64 dense 300 2
    timeout 600 python tools/tile2_check.py 8 16384 64 diag 300
  done; } 2>&1 | grep -v amdgpu.ids | tee $out/tile2_check.txt
