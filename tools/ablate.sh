#!/bin/bash
# timing experiment: run quick_bench under rocprofv3 for each ablated build (tools only, not product)
export PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" 1 2 4 8 15; do
  if [ -z "$v" ]; then unset HENS_LIB; else export HENS_LIB=$R/eryn_amd/lib/ablate_$v.so; fi
  rm -rf /tmp/pa; rocprofv3 --kernel-trace -d /tmp/pa -o qb -- python $R/tools/quick_bench.py --steps 600 "$@" > /tmp/pa.log 2>&1
  echo "== ablate=$v $(grep walker-steps /tmp/pa.log | head -1)"
  python $R/tools/prof_summary.py /tmp/pa/qb_results.db | grep -E "k_stretch|k_pt_cascade|k_plan|k_adapt"
done
