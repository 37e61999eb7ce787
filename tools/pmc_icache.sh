#!/bin/bash
# instruction-cache requests / misses of the two stepping launches, the shape alone and as a lone pipeline rank:
#   bash tools/pmc_icache.sh T W D
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ic; rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS --output-format csv -d /tmp/ic -o q -- python $R/tools/pipe_prof.py $1 $2 $3 60 > /dev/null 2>/tmp/ic_err.log
python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/ic/**/*counter_collection.csv', recursive=True)
if not f:
    print(open('/tmp/ic_err.log').read()[-600:]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name']
    if 'k_stretch_fast' in k or 'k_split1_pt' in k:
        agg[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(f"{k:70s} launches {len(next(iter(c.values()))):5d}  " + "  ".join(f"{n} {v:12.0f}" for n, v in sorted(m.items())) + (f"  miss rate {m.get('SQC_ICACHE_MISSES', 0) / max(m.get('SQC_ICACHE_REQ', 1), 1):.3f}" if 'SQC_ICACHE_REQ' in m else ""))
PY
