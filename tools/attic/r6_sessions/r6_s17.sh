#!/bin/bash
# round 6, session 17: a rank's first launch - k_stretch_fast<PIPE> / k_stretch2<PIPE> with the lead's chain in front of the barrier /
# in the gathers' shadow (library B); three alternations, best of five blocks each
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06p; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 1500 python -m pytest tests/test_hip_pipeline.py -q -x -k "persistent_pipelined" > $out/pytest_pipe2.txt 2>&1; tail -5 $out/pytest_pipe2.txt
for rep in 1 2 3; do
  echo "== fast";   HENS_NO_TILE2_PIPE=1 python tools/pipe_prof.py 8 16384 64 400 2>&1 | grep "^pipe"
  echo "== tile2 front"; python tools/pipe_prof.py 8 16384 64 400 2>&1 | grep "^pipe"
  echo "== tile2 shadow"; HENS_LIB=$R/eryn_amd/lib/libhens_b.so python tools/pipe_prof.py 8 16384 64 400 2>&1 | grep "^pipe"
done | tee $out/pipe_ab.txt
