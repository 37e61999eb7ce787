#!/bin/bash
# A pipeline rank's launches on the AQL queue (default) against the HIP stream (HENS_PIPE_NO_AQL=1): three alternations, then the
# pipeline tests
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2 3; do for l in hip aql; do
  if [ $l = hip ]; then export HENS_PIPE_NO_AQL=1; else unset HENS_PIPE_NO_AQL; fi
  for shape in "16 4096 32 400" "8 16384 64 200" "4 8192 128 200"; do echo -n "$l "; python tools/pipe_prof.py $shape 2>&1 | grep "^pipe" | cut -c1-60; done
done; done
unset HENS_PIPE_NO_AQL
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest tests/test_hip_pipeline.py tests/test_hip_fullsize.py -x -q -m gpu 2>&1 | tail -6 | cut -c1-200; fi
