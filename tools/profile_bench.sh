#!/bin/bash
# Profile the contract bench under rocprofv3 (run on the GPU box via gpurun):
#   bash tools/profile_bench.sh <tag> [bench args]
# writes gpurun_out/<tag>/{kernel_stats.csv, kernel_summary.txt, pmc_fetch.txt, pmc_write.txt}
set -u
TAG=${1:-r01}; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 500 --warmup 100 --no-cpu $*"
# 1) kernel trace + stats (per-kernel time)
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o kt -- python $R/bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stderr.log
cp $(find /tmp/p1 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
python - <<PY > $OUT/kernel_summary.txt
import csv, glob, collections
f = glob.glob('/tmp/p1/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
print("command: rocprofv3 --kernel-trace --stats -- python bench.py $ARGS")
print(f"{'kernel':80s} {'calls':>7s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'pct':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:80]:80s} {len(v):7d} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:9.2f} {max(v)/1e3:9.2f} {sum(v)/1e6:9.2f} {100*sum(v)/tot:6.1f}")
PY
# 2) HBM traffic counters, one pass each (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p2; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/p2 -o pmc -- python $R/bench.py --steps 100 --warmup 20 --no-cpu $* > /dev/null 2>> $OUT/rocprof_stderr.log
  python - <<PY > $OUT/pmc_$C.txt
import csv, glob, collections
f = glob.glob('/tmp/p2/**/*counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(list)
for r in rows:
    if r['Counter_Name'] == '$C':
        agg[r['Kernel_Name']].append(float(r['Counter_Value']))
print("counter $C (KB per dispatch as reported by rocprofv3; gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x)")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:80]:80s} dispatches {len(v):6d}  mean {sum(v)/len(v):14.1f}  min {min(v):12.1f}  max {max(v):12.1f}")
PY
done
ls -la $OUT
