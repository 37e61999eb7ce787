#!/bin/bash
# config 2 on the driver's flags and in long blocks: base library vs the shipped one, four alternations
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2 3 4; do for l in base new; do
  if [ $l = base ]; then export HENS_LIB=$R/ab_live/libhens_base.so; else unset HENS_LIB; fi
  echo -n "$l K=20: "; python bench.py --steps 20 --warmup 5 --no-cpu --no-other 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,3), [round(k['avg_launch_us'],2) for k in d['roofline']['kernels']])"
  echo -n "$l long: "; python bench.py --no-cpu --no-other 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,3))"
done; done
