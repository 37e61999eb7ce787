#!/bin/bash
# round 6, session 8: config 4 - rung dispatch order probe; counters available; RJ profile with instruction classes; bench under rocprofv3
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06h; mkdir -p $out; cd $R; export PYTHONPATH=$R
for rep in 1 2 3; do
  for rr in 0 1; do
    echo -n "HENS_RJ_RUNG_REV=$rr: "; HENS_RJ_RUNG_REV=$rr timeout 300 python bench.py --workload cfg4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), 'us/iter', d['config']['mean_active_leaves_per_walker'])"
  done
done 2>&1 | tee $out/rung_rev.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee $out/leaves_per_rung.txt
import numpy as np, bench, argparse
from eryn_amd.moves.tempering import make_ladder
# leaf statistics per rung after the bench's warm-up: where do the heavy walkers sit?
import bench as b
a = argparse.Namespace(ntemps=None, nwalkers=None, ndim=None, steps=50, warmup=100)
b.BLOCKS = 5
import eryn_amd.rj as rj
orig = rj.RJEngine.close
def close(self):
    x, inds, L, P, betas = self.download()
    n = sum(v.sum(axis=-1) for v in inds.values())
    print("mean leaves per rung:", np.round(n.mean(axis=1), 2), " std within rung:", np.round(n.std(axis=1), 2), " max:", n.max(axis=1))
    orig(self)
rj.RJEngine.close = close
b.run_cfg4(a)
PY
cd /tmp && export TMPDIR=/tmp; rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_INSTS_VALU[A-Z0-9_]*|SQ_INSTS_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $out/sq_counters.txt; cat $out/sq_counters.txt; echo
cd $R; bash tools/profile_rj.sh r06h_cfg4 > $out/profile_rj.log 2>&1; tail -60 $out/profile_rj.log | head -80
bash tools/profile_bench.sh r06h > $out/profile.log 2>&1; cat $R/gpurun_out/r06h/kernel_summary.txt | head -8; head -c 600 $R/gpurun_out/r06h/bench_under_rocprof.json; echo; tail -3 $R/gpurun_out/r06h/rocprof_stderr.log
