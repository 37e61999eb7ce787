#!/bin/bash
# A/B of a pipeline rank's launches: base library vs the shipped one, three alternations; optionally the pipeline tests
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2 3; do for l in base new; do
  if [ $l = base ]; then export HENS_LIB=$R/ab_live/libhens_base.so; else unset HENS_LIB; fi
  for shape in ${SHAPES:-"16 4096 32 400" "4 8192 128 200"}; do echo -n "$l "; python tools/pipe_prof.py $shape 2>&1 | grep "^pipe" | cut -c1-140; done
done; done
unset HENS_LIB
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest tests/test_hip_pipeline.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-200; fi
