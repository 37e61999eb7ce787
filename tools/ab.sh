#!/bin/bash
# A/B timing of env knobs under rocprofv3 (tools only)
export PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  rm -rf /tmp/pa; env $cfg rocprofv3 --kernel-trace -d /tmp/pa -o qb -- python $R/tools/quick_bench.py --steps 600 $QB_ARGS > /tmp/pa.log 2>&1
  echo "== $cfg $(grep walker-steps /tmp/pa.log | head -1)"
  python $R/tools/prof_summary.py /tmp/pa/qb_results.db | grep -E "k_stretch|k_pt_cascade|k_plan|k_adapt"
done
