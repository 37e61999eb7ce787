#!/bin/bash
# rocprofv3 kernel statistics of an arbitrary python command: bash tools/kernel_stats.sh tools/mh_bench.py [args]
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pks -o k -- python $R/"$@" > /tmp/pks_out.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/pks/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-72s calls=%6s avg=%8.2f min=%8.2f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
