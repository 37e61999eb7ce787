#!/bin/bash
# D = 32 dense shapes, base library vs the shipped one (matrix-pipe likelihood at D = 32): one-launch iteration, short tiles, padded rows
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2 3; do for l in base new; do
  if [ $l = base ]; then export HENS_LIB=$R/ab_live/libhens_base.so; else unset HENS_LIB; fi
  for shape in "8 4096 32" "10 4096 32" "16 4096 20" "4 2048 32"; do set -- $shape; echo -n "$l $shape: "; python bench.py --ntemps $1 --nwalkers $2 --ndim $3 --no-cpu --no-other 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,3), [round(k['avg_launch_us'],2) for k in d['roofline']['kernels']])"; done
done; done
