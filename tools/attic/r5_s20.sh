#!/bin/bash
# k_split1_pt: the matrix operand of phase C requested in front of the barrier - 8 x 16384 x 64 (and config 5's dense cousin 4 x 8192 x 128)
# alone and as a rank, base vs new, three alternations; the second launch's phase stamps
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2 3; do for l in base new; do
  if [ $l = base ]; then export HENS_LIB=$R/ab_live/libhens_base.so; else unset HENS_LIB; fi
  for shape in "8 16384 64 300" "4 8192 128 300" "16 4096 32 400"; do echo -n "$l "; python tools/pipe_prof.py $shape 2>&1 | grep "^pipe\|^single" | cut -c1-130 | tr '\n' '|'; echo; done
done; done
unset HENS_LIB
python tools/trace_pipe_phases.py 8 16384 64 2>&1 | grep "second.*single" | cut -c1-200
