#!/bin/bash
# config 4 (k_rj) under rocprofv3: kernel trace + one SQ counter pass -> gpurun_out/<tag>/{kernel_summary.txt, rj_valu.json}
#   VALU lane-instructions per launch (SQ_INSTS_VALU x 64) against the chip's issue rate: the compute roofline of a
#   kernel that is bound by FP64 transcendentals, not by HBM (bench.py --workload cfg4 reads profiles/rj_valu.json)
TAG=${1:-r03_cfg4}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r1 -o kt -- python $R/bench.py --workload cfg4 --steps 100 --warmup 20 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/err.log
rm -rf /tmp/r2; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/r2 -o q -- python $R/bench.py --workload cfg4 --steps 30 --warmup 10 --no-cpu > /dev/null 2>> $OUT/err.log
# round 6: the launch's arithmetic by instruction class (FP64 add / mul / fma / transcendental, integer, conversions), a pass of its own
rm -rf /tmp/r3; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU --output-format csv -d /tmp/r3 -o m -- python $R/bench.py --workload cfg4 --steps 30 --warmup 10 --no-cpu > /dev/null 2>> $OUT/err.log
rm -rf /tmp/r4; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d /tmp/r4 -o n -- python $R/bench.py --workload cfg4 --steps 30 --warmup 10 --no-cpu > /dev/null 2>> $OUT/err.log
python - <<PY
import csv, glob, collections, json
rows = list(csv.DictReader(open(glob.glob('/tmp/r1/**/*kernel_trace.csv', recursive=True)[0])))
agg = collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
with open('$OUT/kernel_summary.txt', 'w') as f:
    f.write("command: rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg4 --steps 100 --warmup 20 --no-cpu\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write(f"{k[:80]:80s} calls {len(v):6d} avg_us {sum(v)/len(v)/1e3:9.2f} min_us {min(v)/1e3:9.2f} total_ms {sum(v)/1e6:9.2f} pct {100*sum(v)/tot:5.1f}\n")
rj_all = [d for k, v in agg.items() if 'k_rj' in k for d in v]      # (every instantiation of k_rj: the counters below average over the same launches)
rj_us = [sum(rj_all) / len(rj_all) / 1e3] if rj_all else []
rows = list(csv.DictReader(open(glob.glob('/tmp/r2/**/*counter_collection.csv', recursive=True)[0])))
c = collections.defaultdict(list)
for r in rows:
    if 'k_rj' in r['Kernel_Name']:
        c[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in c.items()}
out = {"source": "rocprofv3 --pmc SQ_* on bench.py --workload cfg4 (tools/profile_rj.sh)", "k_rj_avg_us": rj_us[0] if rj_us else None,
       "counters_per_launch": m, "valu_lane_insts_per_launch": m.get("SQ_INSTS_VALU", 0) * 64}
# per instantiation: durations from the kernel trace, the instruction mix from the class counters (per launch, per wave = per walker)
def short(n):
    import re
    m = re.search(r'k_rj<[^>]*>', n)
    return m.group(0) if m else None
dur = collections.defaultdict(list)
for k, v in agg.items():
    if short(k): dur[short(k)] += v
mix = collections.defaultdict(lambda: collections.defaultdict(list))
for d_ in ('/tmp/r2', '/tmp/r3', '/tmp/r4'):
    for f_ in glob.glob(d_ + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f_)):
            if short(r['Kernel_Name']): mix[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
per = {}
for k, cs in mix.items():
    m_ = {n: sum(v) / len(v) for n, v in cs.items()}
    waves = m_.get('SQ_WAVES', 16384.0)
    us = sum(dur[k]) / len(dur[k]) / 1e3 if dur.get(k) else None
    f64 = {n: m_.get('SQ_INSTS_VALU_' + n + '_F64', 0.0) for n in ('ADD', 'MUL', 'FMA', 'TRANS')}
    flops = 64.0 * (f64['ADD'] + f64['MUL'] + 2.0 * f64['FMA'] + f64['TRANS'])
    per[k] = {"avg_us": us, "waves": waves, "per_wave": {n: v / waves for n, v in m_.items() if n.startswith('SQ_INSTS')},
              "fp64_flops_per_launch": flops, "fp64_TFLOPs": None if not us else flops / (us * 1e-6) / 1e12,
              "frac_of_fp64_vector_peak_78.6TF": None if not us else flops / (us * 1e-6) / 78.6e12,
              "valu_issue_frac": None if not us else m_.get('SQ_INSTS_VALU', 0.0) * 64 / (us * 1e-6) / (256 * 4 * 16 * 2.4e9)}
out["per_instantiation"] = per
json.dump(out, open('$OUT/rj_valu.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
cat $OUT/kernel_summary.txt
