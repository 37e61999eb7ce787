#!/bin/bash
# DEV PROBE: config 5 on one GPU with the MH launch's Philox calls replaced by nothing (timing only, wrong values)
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2; do for l in new fake; do
  if [ $l = fake ]; then export HENS_LIB=$R/ab_live/libhens_fake.so; else unset HENS_LIB; fi
  echo -n "$l cfg5 one GPU: "; python bench.py --workload cfg5 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), 'us', [round(k['avg_launch_us'],1) for k in d['roofline']['kernels']], d['config']['gaussian_acceptance'])"
done; done
