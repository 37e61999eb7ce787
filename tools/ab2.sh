#!/bin/bash
# A/B of env knobs: plain timing + phase trace (tools only)
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python tools/quick_bench.py --steps 2000 | head -1
  env $cfg python tools/trace_phases.py | tail -2 | head -1
done
