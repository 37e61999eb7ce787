#!/bin/bash
# phase E with its LDS reads batched: single-GPU shapes base vs new, three alternations (long blocks on the AQL queue)
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2 3; do for l in base new; do
  if [ $l = base ]; then export HENS_LIB=$R/ab_live/libhens_base.so; else unset HENS_LIB; fi
  for shape in "16 4096 32" "8 16384 64" "4 8192 128"; do set -- $shape; echo -n "$l $shape: "; python bench.py --ntemps $1 --nwalkers $2 --ndim $3 --no-cpu --no-other 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,3), [round(k['avg_launch_us'],2) for k in d['roofline']['kernels']])"; done
done; done
