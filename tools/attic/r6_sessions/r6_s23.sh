#!/bin/bash
# round 6, session 23: bench.py's live PMC traffic (two rocprofv3 passes over a child process) beside the static file; the long fence-free
# guard on the config-3 shard (k_stretch2)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06t; mkdir -p $out; cd $R; export PYTHONPATH=$R
( time python bench.py --steps 20 --warmup 5 --no-other > $out/b_live.json 2> $out/err.log ) 2>&1 | tail -3
( time python bench.py --steps 20 --warmup 5 --no-other --no-live-traffic > $out/b_static.json 2>> $out/err.log ) 2>&1 | tail -3
python - <<'PY'
import json
for f in ("b_live", "b_static"):
    d = json.load(open(f"gpurun_out/r06t/{f}.json")); r = d["roofline"]
    print(f, "us %.3f" % (d["ms_per_step"] * 1e3), "traffic", r["traffic"], "frac_traffic %.3f" % r["frac_traffic"], [(k["kernel"].split()[0], k["traffic"]) for k in r["kernels"]])
    print("   ", r.get("traffic_source"), r.get("traffic_live_unavailable"))
PY
tail -5 $out/err.log
timeout 1200 python -m pytest tests/test_hip_records.py -q -k "2000_iterations and 16384" 2>&1 | tail -3
