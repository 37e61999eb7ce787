#!/bin/bash
# Same-box A/B (see ab_lib.sh): per-kernel durations and the gaps in front of them from a rocprofv3 kernel trace
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
B=$R/ab_live/libhens_base.so
cd /tmp && export TMPDIR=/tmp
for which in base new base new; do
  rm -rf /tmp/p3
  if [ $which = base ]; then export HENS_LIB=$B; else unset HENS_LIB; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -o kt -- python $R/tools/quick_bench.py --steps 1000 > /tmp/p3.log 2>&1
  echo "== $which $(grep walker-steps /tmp/p3.log | head -1 | cut -c60-)"
  python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/p3/**/*kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'k_plan' not in r['Kernel_Name']]
d = collections.defaultdict(list); gaps = collections.defaultdict(list); prev = None
for name, s, e in seq[len(seq)//2:]:
    d[name].append(e - s)
    if prev: gaps[name].append(s - prev[2])
    prev = (name, s, e)
for k in d:
    v = sorted(d[k]); g = sorted(gaps[k]) or [0]
    if len(v) > 50: print(f"  {k:20s} n={len(v):5d} dur avg {sum(v)/len(v)/1e3:6.2f} med {v[len(v)//2]/1e3:6.2f}   gap-before avg {sum(g)/len(g)/1e3:5.2f} med {g[len(g)//2]/1e3:5.2f}")
PY
done
