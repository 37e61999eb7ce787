#!/bin/bash
# round 6, session 13: k_stretch2 tests, the full suite, the bench line, the config-3 shard profile with the new kernel
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06r; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 1500 python -m pytest tests/test_hip_records.py -m gpu -x -q -k "persistent_pipelined" > $out/pytest_tile2.txt 2>&1; tail -6 $out/pytest_tile2.txt
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; tail -6 $out/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench.err; tail -2 $out/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06r/bench_steps20.json"))
print("cfg2", d["ms_per_step"] * 1e3, [k["avg_launch_us"] for k in d["roofline"]["kernels"]])
o = d["other_shapes"]["config_3_shard"]; print("cfg3 shard", o["ms_per_step"] * 1e3, [(k["kernel"][:14], k["avg_launch_us"], k["frac"]) for k in o["kernels"]], o["whole_path_frac"], o["sum_kernel_us_per_iteration"])
PY
bash tools/profile_bench.sh r06r_cfg3shard --ntemps 8 --nwalkers 16384 --ndim 64 > $out/profile_cfg3.log 2>&1; head -6 $R/gpurun_out/r06r_cfg3shard/kernel_summary.txt; head -4 $R/gpurun_out/r06r_cfg3shard/pmc_FETCH_SIZE.txt
