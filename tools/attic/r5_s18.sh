#!/bin/bash
# k_pt_cascade with a thread per element and straight-line walks: config 4 / config 5 base vs new, phase stamps, then the suite
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2; do for l in base new; do
  if [ $l = base ]; then export HENS_LIB=$R/ab_live/libhens_base.so; else unset HENS_LIB; fi
  echo -n "$l cfg5: "; python bench.py --workload cfg5 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), 'us', [round(k['avg_launch_us'],1) for k in d['roofline']['kernels']])"
  echo -n "$l cfg4: "; python bench.py --workload cfg4 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), 'us')"
done; done
unset HENS_LIB
HENS_NO_FUSED=1 python tools/trace_pt2.py 32 8192 32 2>&1 | grep k_pt; HENS_NO_FUSED=1 python tools/trace_pt2.py 8 2048 32 2>&1 | grep k_pt
timeout 2000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^  [0-9]" | tail -4 | cut -c1-200
