#!/bin/bash
# DEV PROBE (timing only): config 4 with a pulse's exp replaced by two FP64 operations - the ceiling of any cheaper-exp scheme
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for l in rjreal rjfake; do
  HENS_LIB=$R/ab_live/libhens_$l.so python bench.py --workload cfg4 --no-cpu 2>&1 | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$l', d['ms_per_step'] * 1e3, 'us/iter; leaves', d['config']['mean_active_leaves_per_walker'], 'acc', d['config']['accept_in_model'], d['config']['accept_birth_death'])"
done
