#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
export HENS_AQL_STATS=1
timeout 600 python tools/aql_check.py > $O/aql_check.txt 2>&1; echo "rc=$?" >> $O/aql_check.txt
timeout 120 python tools/short_call.py > $O/short_aql.log 2>&1
timeout 900 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench20.json 2> $O/bench20.err
python bench.py --no-cpu > $O/bench_long.json 2> $O/bench_long.err
grep -v "memory pool" $O/aql_check.txt | grep -v amdgpu.ids; cat $O/short_aql.log | grep -v "memory pool"; tail -n 5 $O/suite.log; python -c "
import json
for f in ['bench20','bench_long']:
    d=json.load(open('$O/'+f+'.json')); print(f, d['ms_per_step'], d['value'], d['block_ms'], d['roofline']['whole_path_frac'])
"
