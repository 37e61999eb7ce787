#!/bin/bash
# Same-box A/B of two builds: ab_live/libhens_base.so (HENS_LIB=... python -m eryn_amd._build from the older sources)
# against the current library: throughput, per-launch events, in-kernel phase stamps of both launches.
export PYTHONPATH=$GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/ab_live/libhens_base.so
for i in 1 2 3; do
  echo -n "base: "; HENS_LIB=$B timeout 120 python tools/quick_bench.py --prof 0 2>&1 | head -1 | cut -c60-
  echo -n "new:  "; timeout 120 python tools/quick_bench.py --prof 0 2>&1 | head -1 | cut -c60-
done
echo -n "base: "; HENS_LIB=$B timeout 120 python tools/quick_bench.py --prof 1 2>&1 | sed -n 3,3p
echo -n "new:  "; timeout 120 python tools/quick_bench.py --prof 1 2>&1 | sed -n 3,3p
echo -n "base s0: "; HENS_LIB=$B timeout 100 python tools/trace_fused.py 16 4096 32 1 2>&1 | grep "phase durations"
echo -n "new  s0: "; timeout 100 python tools/trace_fused.py 16 4096 32 1 2>&1 | grep "phase durations"
echo -n "base fu: "; HENS_LIB=$B timeout 100 python tools/trace_fused.py 2>&1 | grep "phase durations"
echo -n "new  fu: "; timeout 100 python tools/trace_fused.py 2>&1 | grep "phase durations"
