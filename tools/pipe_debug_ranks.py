"""Debug: N in-process pipeline ranks, per-rank ladders / swap counters after every call (which rank parts ways, and when)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HENS_PIPE_TIMEOUT_S", "20")
sys.argv = [sys.argv[0]] + sys.argv[1:]
import tests.pipeline_worker as pw
from eryn_amd.ladder import LadderPipeline, rung_partition
nranks, T, W, D, iters, chunk = map(int, sys.argv[1:7])
_, bounds = rung_partition(T, nranks)
engs = [pw.make(T, W, D, b) for b in bounds]
LadderPipeline.connect_local(engs)
done = 0
while done < iters:
    n = min(chunk, iters - done)
    for e in engs:
        e.step(n)
    for e in engs:
        e.synchronize()
    done += n
    bs = [e.download()[3] for e in engs]
    cs = [e.counters() for e in engs]
    same_b = all(np.array_equal(b, bs[0]) for b in bs)
    same_s = all(np.array_equal(c["swaps_total"], cs[0]["swaps_total"]) for c in cs)
    print(f"after {done:3d} iterations: betas equal {same_b}, swaps_total equal {same_s}")
    if not (same_b and same_s):
        for r, (b, c) in enumerate(zip(bs, cs)):
            print(f"  rank {r}: betas {np.array2string(b, precision=17)}\n          swaps_total {c['swaps_total']} swaps_last {c['swaps_last']} adapt_time {c['adapt_time']}")
        break
