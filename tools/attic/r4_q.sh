#!/bin/bash
O=gpurun_out/r4q; mkdir -p $O
for i in 1 2 3; do
  HENS_LIB=$PWD/build_ab/libhens_pf.so python bench.py --steps 20 --warmup 5 --no-other --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ring pre-touched', round(d['ms_per_step']*1e3,3), [round(x*1e3,1) for x in d['block_ms']])"
  python bench.py --steps 20 --warmup 5 --no-other --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('as built       ', round(d['ms_per_step']*1e3,3), [round(x*1e3,1) for x in d['block_ms']])"
done > $O/pretouch.txt 2>&1
cat $O/pretouch.txt
