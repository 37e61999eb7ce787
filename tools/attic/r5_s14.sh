#!/bin/bash
# A/B of a pipeline rank's launches: base library vs the shipped one, three alternations; optionally the pipeline tests
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2 3; do for l in base new; do
  if [ $l = base ]; then export HENS_LIB=$R/ab_live/libhens_base.so; else unset HENS_LIB; fi
  for shape in "16 4096 32 400" "8 16384 64 200"; do for dl in 0 1; do echo -n "$l "; PIPE_DELAY=$dl python tools/pipe_prof.py $shape 2>&1 | grep "^pipe" | cut -c1-140; done; done
done; done
unset HENS_LIB
PYTHONPATH=$R python tools/trace_pipe_phases.py 16 4096 32 2>&1 | grep "first.*pipe" | cut -c1-330
PYTHONPATH=$R python tools/trace_pipe_phases.py 8 16384 64 2>&1 | grep "first.*pipe" | cut -c1-330
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest tests/test_hip_pipeline.py tests/test_hip_fullsize.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-200; fi
