#!/bin/bash
# Per-split durations of the stretch kernel (split 0 carries the folded ladder adaptation) from a rocprofv3 kernel trace.
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3; rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -o kt -- python $R/bench.py --steps 500 --warmup 100 --no-cpu "$@" > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('/tmp/p3/**/*kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'k_plan' not in r['Kernel_Name']]
import collections
d = collections.defaultdict(list); gaps = collections.defaultdict(list)
prev = None; idx = 0
for name, s, e in seq:
    if 'k_stretch_fast' in name and ', 0, 8' in name:
        key = 'stretch after ' + ('PT' if prev and 'k_pt' in prev[0] else 'stretch' if prev and 'k_stretch' in prev[0] else 'other')
    elif 'k_pt_cascade' in name:
        key = 'pt'
    else:
        key = 'other'
    d[key].append(e - s)
    if prev: gaps[key].append(s - prev[2])
    prev = (name, s, e)
for k in d:
    v = d[k]; g = gaps[k]
    print(f"{k:24s} n={len(v):5d} dur avg {sum(v)/len(v)/1e3:7.2f} min {min(v)/1e3:6.2f}   gap-before avg {sum(g)/max(len(g),1)/1e3:6.2f} min {min(g)/1e3 if g else 0:6.2f}")
PY
