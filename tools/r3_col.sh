#!/bin/bash
# column-ordered records: correctness (oracle replay, record-mode equivalence, sampler) and same-box timing against the
# slot-ordered records (HENS_NO_COL=1) and with the rung's row table staged in LDS (HENS_TAB=1)
out=gpurun_out/${1:-s3o}; mkdir -p $out
export PYTHONPATH=.
timeout 1200 python -m pytest tests/test_hip_replay.py tests/test_hip_records.py tests/test_hip_sampler.py tests/test_hip_rng.py tests/test_production_draws.py -x -q > $out/pytest_col.log 2>&1
tail -8 $out/pytest_col.log
{
for i in 1 2 3; do
  for k in "X=1" "HENS_TAB=1" "HENS_NO_COL=1"; do echo -n "$k: "; env $k python tools/quick_bench.py --steps 4000 --prof 0 2>&1 | grep -o "[0-9.]* us/iter"; done
done
for k in "X=1" "HENS_NO_COL=1"; do echo "== $k"; env $k python tools/quick_bench.py --prof 1 2>&1 | sed -n 2,3p | cut -c1-260; env $k python tools/short_call.py 20 2>&1 | grep "sync | step, eng.sync"; done
for k in "X=1" "HENS_NO_COL=1"; do for m in 1 3; do echo "=== $k mode $m"; env $k python tools/trace_fused.py 16 4096 32 $m 1 2>&1 | grep -E "phase durations|lifetime perc|  bar1 "; done; done
for k in "X=1" "HENS_NO_COL=1"; do echo "== $k"; for shp in "16 16384 32" "8 16384 64" "32 2048 32" "64 1024 16"; do set -- $shp; env $k python tools/quick_bench.py --T $1 --W $2 --D $3 --steps 500 --prof 0 2>&1 | head -1 | cut -c1-110; done; done
} 2>&1 | grep -v amdgpu.ids > $out/time_col.log
cat $out/time_col.log
