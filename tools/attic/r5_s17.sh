#!/bin/bash
# DEV PROBE: config 5 on one GPU with fewer Philox rounds in the MH launch's normal draws (timing only): LIBS="new f3 f7"
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2; do for l in ${LIBS:-new f3 f7}; do
  if [ $l != new ]; then export HENS_LIB=$R/ab_live/libhens_$l.so; else unset HENS_LIB; fi
  echo -n "$l cfg5 one GPU: "; python bench.py --workload cfg5 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), 'us', [round(k['avg_launch_us'],1) for k in d['roofline']['kernels']], round(d['config']['gaussian_acceptance'],3))"
done; done
