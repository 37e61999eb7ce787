#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
export HENS_LIB=$PWD/build_ab/libhens_rj.so
timeout 900 python -m pytest tests/test_hip_rj.py -x -q -m gpu > $O/rj_tests.log 2>&1
python bench.py --workload cfg4 --steps 200 --warmup 20 > $O/cfg4.json 2> $O/cfg4.err
HENS_RJ_NO_TEMPLATES=1 python bench.py --workload cfg4 --steps 200 --warmup 20 > $O/cfg4_base.json 2> $O/cfg4_base.err
tail -n 8 $O/rj_tests.log
python -c "
import json
for f in ['cfg4','cfg4_base']:
    d=json.load(open('$O/'+f+'.json')); print(f, d['ms_per_step'], d['value'], d['config']['accept_in_model'], d['config']['accept_birth_death'], d['config']['mean_active_leaves_per_walker'])
"
