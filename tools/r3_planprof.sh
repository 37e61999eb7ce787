#!/bin/bash
# kernel durations of the plan kernels (serial: alone on the chip; default: beside the stepping kernels)
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
L=${1:-cur}
for mode in serial concurrent; do
  rm -rf /tmp/pp
  if [ $mode == serial ]; then export HENS_PLAN_SERIAL=1; else unset HENS_PLAN_SERIAL; fi
  HENS_LIB=$GRAFT_REPO_ROOT/build_ab/libhens_$L.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o kt -- python $GRAFT_REPO_ROOT/tools/quick_bench.py --steps 1000 > /tmp/pp.log 2>&1
  grep -o "[0-9.]* us/iter" /tmp/pp.log
  python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/pp/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print("$mode")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if len(v) > 5: print(f"  {k[:70]:70s} calls {len(v):6d} avg {sum(v)/len(v)/1e3:9.2f} us  min {min(v)/1e3:8.2f}  total {sum(v)/1e6:8.2f} ms")
PY
done
