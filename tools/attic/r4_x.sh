#!/bin/bash
# round 4: the library built from seven translation units - whole GPU suite, the driver's bench command, AQL stats
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/x_suite.log
for i in 1 2; do python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/x_bench$i.json; done
HENS_AQL_STATS=1 python bench.py --steps 20 --warmup 5 --no-other 2>gpurun_out/x_aql.err | tail -1 > gpurun_out/x_bench3.json
tail -3 gpurun_out/x_suite.log; python - <<'PY'
import json
for i in (1,2,3):
    d=json.loads(open(f'gpurun_out/x_bench{i}.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('warnings'))
PY
grep -i "aql" gpurun_out/x_aql.err | tail -5
