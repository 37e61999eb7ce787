#!/bin/bash
# round 6, session 26: the whole -m gpu suite on the final tree + the r06z bench lines with the final bench.py (live PMC traffic)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06z; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench.err
python bench.py --no-cpu --no-other > $out/bench.json 2>> $out/bench.err
python bench.py --workload cfg4 > $out/bench_cfg4.json 2>> $out/bench.err
python bench.py --workload cfg5 > $out/bench_cfg5.json 2>> $out/bench.err
HENS_DIST_BACKEND=gloo GPU_MAX_HW_QUEUES=16 python bench.py --gpus 2 --ntemps 8 --nwalkers 256 --ndim 32 --steps 20 --warmup 5 --no-cpu > $out/bench_gpus2_dryrun.json 2>> $out/bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06z/bench_steps20.json")); r = d["roofline"]
print("us %.3f" % (d["ms_per_step"] * 1e3), "frac %.3f" % r["frac"], "fit", r["kernels_fit_in_timed_iteration"], "sum %.3f" % r["sum_kernel_us_per_iteration"], "traffic", r["traffic"], "|", (r.get("traffic_source") or "")[:60])
PY
