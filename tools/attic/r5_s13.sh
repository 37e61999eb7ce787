#!/bin/bash
# Round 5, session 13: full GPU suite, the driver's bench command, config 4 line + profile with the shipped library
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5_s13; mkdir -p $out
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 | cut -c1-200 > $out/pytest_gpu.txt; cat $out/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench.err; python - <<PY
import json
d = json.loads(open("$out/bench_steps20.json").read().strip().splitlines()[-1])
print("config 2:", d["ms_per_step"] * 1e3, "us", d["value"], "cold", d["cold_blocks"]["ms_per_step"] * 1e3, "roofline", {k: d["roofline"].get(k) for k in ("frac", "whole_path_frac", "frac_traffic")})
for k, v in d["other_shapes"].items():
    print(k, {kk: v.get(kk) for kk in ("ms_per_step", "value", "error")}, (v.get("roofline") or {}).get("frac"), v.get("whole_path_frac"))
PY
tail -3 $out/bench.err
python bench.py --workload cfg4 > $out/bench_cfg4.json 2>> $out/bench.err; head -c 400 $out/bench_cfg4.json; echo
bash tools/profile_rj.sh r05c_cfg4 2>&1 | tail -12
