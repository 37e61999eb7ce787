#!/bin/bash
# round 6, session 1: the new guard tests + the bench line with dispatch-timestamp rooflines
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06a; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 1500 python -m pytest tests/test_hip_records.py -m gpu -x -q -k "fence_free or dispatch_timestamps" > $out/pytest_new.txt 2>&1; tail -15 $out/pytest_new.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench.err; tail -3 $out/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06a/bench_steps20.json"))
r = d["roofline"]
print("cfg2", d["ms_per_step"] * 1e3, "sum", r["sum_kernel_us_per_iteration"], r["kernels_fit_in_timed_iteration"], "frac", r["frac"], r["launch_clock"][:20], "span", r["span_us_per_iteration"])
for k in r["kernels"]: print("   ", k["kernel"][:16], k["avg_launch_us"], k["frac"])
for n, o in d["other_shapes"].items():
    rr = o.get("roofline", o)
    print(n, o.get("ms_per_step", 0) * 1e3, "sum", rr.get("sum_kernel_us_per_iteration"), rr.get("kernels_fit_in_timed_iteration"), "exceeds", rr.get("accounting_exceeds_traffic"), "wp", rr.get("whole_path_frac"), (rr.get("launch_clock") or "")[:12])
    for k in rr.get("kernels", []): print("   ", k["kernel"][:16], k["avg_launch_us"], k.get("frac"), k.get("frac_kind"), k.get("frac_8d_flagged"))
PY
