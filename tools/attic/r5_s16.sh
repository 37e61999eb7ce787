#!/bin/bash
# four MH normals per Philox call: config 5 (one GPU and a shard) base vs new, then the MH / replay / pipeline tests
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2; do for l in base new; do
  if [ $l = base ]; then export HENS_LIB=$R/ab_live/libhens_base.so; else unset HENS_LIB; fi
  echo -n "$l cfg5 one GPU: "; python bench.py --workload cfg5 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), 'us', [round(k['avg_launch_us'],1) for k in d['roofline']['kernels']], d['config']['gaussian_acceptance'])"
done; done
unset HENS_LIB
timeout 1500 python -m pytest tests/test_hip_mh.py tests/test_hip_replay.py tests/test_hip_sampler.py tests/test_hip_rj.py -x -q -m gpu 2>&1 | tail -5 | cut -c1-200
