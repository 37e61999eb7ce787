#!/bin/bash
# SQ stall breakdown of the stepping kernels (one rocprofv3 --pmc pass, 8 SQ slots): bash tools/pmc_sq.sh T W D
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for CS in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
          "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES"; do
rm -rf /tmp/pq; rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pq -o q -- python $R/tools/quick_bench.py --T $1 --W $2 --D $3 --steps 60 --warmup 20 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name']
    if ('k_stretch_fast' in k and ', 0, 8' in k) or 'k_pt_cascade' in k or 'k_split1_pt' in k or 'k_iter' in k:
        agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"    {c:24s} {sum(v)/len(v):16.0f}")
PY
done
