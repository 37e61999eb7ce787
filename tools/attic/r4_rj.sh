#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "rj" 2>&1 | tail -4
for i in 1 2; do python bench.py --workload cfg4 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/rj_bench$i.json; python -c "
import json; d=json.load(open('gpurun_out/rj_bench$i.json')); print(d['value'], d['ms_per_step'], d['config']['accept_in_model'], d['config']['accept_birth_death'], d['config']['mean_active_leaves_per_walker'])"; done
