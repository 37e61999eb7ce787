#!/bin/bash
python -m pytest tests -m gpu -x -q -k "rj" 2>&1 | tail -3
for rep in 1 2; do for f in 0 1; do
  echo -n "HENS_NO_FOLD=$f: "; ( [ $f = 1 ] && export HENS_NO_FOLD=1; python bench.py --workload cfg4 --steps 20 --warmup 5 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us/iter  %.4g  acc %.9f %.9f' % (d['ms_per_step']*1e3, d['value'], d['config']['accept_in_model'], d['config']['accept_birth_death']))" )
done; done
