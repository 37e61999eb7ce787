#!/bin/bash
# the one-launch iteration (k_iter) forced onto config 2 vs the two launches, same box
out=gpurun_out/${1:-s3c}; mkdir -p $out
export PYTHONPATH=.
{
echo "== k_iter forced (HENS_ITER_MAX=2)"; HENS_ITER_MAX=2 python tools/quick_bench.py --prof 0 | head -1
HENS_ITER_MAX=2 python tools/quick_bench.py --prof 1 | head -3
echo "== two launches"; python tools/quick_bench.py --prof 0 | head -1
python tools/quick_bench.py --prof 1 | head -3
} > $out/iter_cfg2.log 2>&1
cat $out/iter_cfg2.log
