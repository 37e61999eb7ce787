#!/bin/bash
# same-box A/B of k_rj builds at config 4: tools/r4_rjab.sh name1 name2 ... (build_ab/libhens_<name>.so)
for rep in 1 2; do for L in "$@"; do
  echo -n "$L: "; HENS_LIB=$PWD/build_ab/libhens_$L.so python bench.py --workload cfg4 --steps 20 --warmup 5 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us/iter  %.4g  acc %.6f %.6f' % (d['ms_per_step']*1e3, d['value'], d['config']['accept_in_model'], d['config']['accept_birth_death']))"
done; done
