#!/bin/bash
# Same-box A/B of the one-launch iteration: base library (tools/mkbase.sh) vs current, config 2 (+ replay test of the current one)
export PYTHONPATH=$GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/ab_live/libhens_base.so
timeout 300 python -m pytest tests/test_hip_replay.py -x -q -m gpu -k "config2 or small_two or diag" 2>&1 | tail -2
for i in 1 2; do
  echo -n "base: "; HENS_LIB=$B timeout 120 python tools/quick_bench.py --prof 0 2>&1 | head -1 | cut -c60-
  echo -n "new:  "; timeout 120 python tools/quick_bench.py --prof 0 2>&1 | head -1 | cut -c60-
done
echo -n "base: "; HENS_LIB=$B timeout 120 python tools/quick_bench.py --prof 1 2>&1 | sed -n 2,2p
echo -n "new:  "; timeout 120 python tools/quick_bench.py --prof 1 2>&1 | sed -n 2,2p
echo -n "base: "; HENS_LIB=$B timeout 100 python tools/trace_fused.py 16 4096 32 3 1 2>&1 | grep "phase durations\|lifetime perc"
echo -n "new:  "; timeout 100 python tools/trace_fused.py 16 4096 32 3 1 2>&1 | grep "phase durations\|lifetime perc"
