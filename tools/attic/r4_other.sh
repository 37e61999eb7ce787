#!/bin/bash
# the N = 1 bench line's other_shapes with build_ab libs: tools/r4_other.sh name ... ("main" = in-tree)
for rep in 1 2; do for L in "$@"; do
  if [ $L = main ]; then unset HENS_LIB; else export HENS_LIB=$PWD/build_ab/libhens_$L.so; fi
  echo "$L: "; python bench.py --steps 20 --warmup 5 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('   cfg2 %.2f us' % (d['ms_per_step']*1e3))
for k,v in d['other_shapes'].items(): print('  ', k, '%.2f us/iter' % (v['ms_per_step']*1e3), [(q['kernel'][:14], round(q['avg_launch_us'],2)) for q in v['kernels']])"
done; done
