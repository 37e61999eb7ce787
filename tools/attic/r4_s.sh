#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1
tail -n 4 $O/suite.log
python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
python bench.py --no-cpu --no-other > $O/bench_long.json 2> $O/bench_long.err
python tools/block_series.py 80 2>&1 | grep -v amdgpu > $O/block_series.txt
python -c "
import json
d=json.load(open('$O/bench20.json')); print('steps20', d['ms_per_step'], d['value'], d['timing'], d['cold_blocks'], d['roofline']['whole_path_frac'])
for k,v in d['other_shapes'].items(): print(k, v['ms_per_step'], v['cold_blocks']['ms_per_step'], v['whole_path_frac'])
d=json.load(open('$O/bench_long.json')); print('long', d['ms_per_step'], d['value'], d['timing'], d['roofline']['whole_path_frac'])
"
