#!/bin/bash
# round 6, session 24: what rocprofv3 puts into its child's environment (live_traffic's "am I profiled" test); bench under rocprofv3 without --no-cpu
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06t; mkdir -p $out; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
env | grep -i "rocp\|preload" | head; echo "== under rocprofv3:"
rocprofv3 --kernel-trace -d /tmp/pp -o x -- env 2>/dev/null | grep -i "rocp\|preload" | cut -c1-200 | head
echo "== bench under rocprofv3 (default secondary figures on):"
( time rocprofv3 --kernel-trace --stats -d /tmp/pq -o y -- python $R/bench.py --steps 20 --warmup 5 --no-other --cpu-seconds 2 > $out/b_nested.json 2> $out/nested.err ) 2>&1 | tail -3
python - <<PY
import json
d = json.load(open("$out/b_nested.json")); r = d["roofline"]
print("us %.3f" % (d["ms_per_step"] * 1e3), r.get("traffic_source"), "|", r.get("traffic_live_unavailable"))
PY
