#!/bin/bash
# round 6, session 19: k_stretch2<PIPE> on the delayed schedule; rank timings both schedules, three alternations
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06p; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 2400 python -m pytest tests/test_hip_pipeline.py -q -k "persistent_pipelined or delayed" > $out/pytest_pipe3.txt 2>&1; tail -8 $out/pytest_pipe3.txt
for rep in 1 2 3; do
  for d in 0 1; do
  echo "== fast delay=$d";   PIPE_DELAY=$d HENS_NO_TILE2_PIPE=1 python tools/pipe_prof.py 8 16384 64 400 2>&1 | grep "^pipe\|^single"
  echo "== tile2 delay=$d"; PIPE_DELAY=$d python tools/pipe_prof.py 8 16384 64 400 2>&1 | grep "^pipe"
  done
done | tee $out/pipe_ab2.txt
