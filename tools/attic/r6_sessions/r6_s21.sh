#!/bin/bash
# round 6, session 21: the bench line's consistency check with the profiled pass's middle-half statistic - four driver-style runs + the long blocks
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06q; mkdir -p $out; cd $R; export PYTHONPATH=$R
for i in 1 2 3 4; do python bench.py --steps 20 --warmup 5 --no-other --no-cpu > $out/b20_$i.json 2>> $out/err.log; done
python bench.py --no-cpu --no-other > $out/blong.json 2>> $out/err.log
python bench.py --steps 20 --warmup 5 > $out/bfull.json 2>> $out/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06q/b*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], "us %.3f" % (d["ms_per_step"] * 1e3), "sum %.3f" % r["sum_kernel_us_per_iteration"], "fit", r["kernels_fit_in_timed_iteration"], "prof/timed %.4f" % r.get("profiled_over_timed", 0), "frac %.3f" % r["frac"], [round(k["avg_launch_us"], 2) for k in r["kernels"]])
    for k, e in (d.get("other_shapes") or {}).items():
        rr = e.get("roofline", e)
        if "sum_kernel_us_per_iteration" in rr: print("   ", k, "us %.2f" % (e["ms_per_step"] * 1e3), "sum %.2f" % rr["sum_kernel_us_per_iteration"], "fit", rr["kernels_fit_in_timed_iteration"], "prof/timed %.4f" % rr.get("profiled_over_timed", 0))
PY
