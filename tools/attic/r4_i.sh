#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
build_ab/mfma_f64_rate > $O/mfma_f64_rate.txt 2>&1
timeout 900 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
bash tools/profile_bench.sh r4i_prof > $O/prof.log 2>&1
cat $O/mfma_f64_rate.txt; tail -n 16 $O/suite.log; cat gpurun_out/r4i_prof/kernel_summary.txt | cut -c1-160 | head -8
python -c "
import json
d=json.load(open('$O/bench20.json')); print(d['ms_per_step'], d['value'], d['block_ms']); r=d['roofline']; print({k:v for k,v in r.items() if k not in ('kernels','bytes','traffic_source')})
for k,v in d['other_shapes'].items(): print(k, v['ms_per_step'], v['value'], v['whole_path_frac'], [(kk['kernel'][:14], round(kk['avg_launch_us'],2), round(kk['frac'],3)) for kk in v['kernels']])
"
