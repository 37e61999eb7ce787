"""Long runs of the ladder pipeline (N contexts of ONE process on one GPU, stepped in short alternating calls: one host thread
cannot queue thousands of iterations for one rank before the other's are queued) against one context holding the whole ladder:
bit-identical state and counters.   python tools/soak_pipeline.py"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16"); os.environ.setdefault("HENS_PIPE_TIMEOUT_S", "20")
import numpy as np
KEYS = ("x", "L", "P", "betas", "accepted", "swaps_total", "swaps_last")
cases = ((2, 8, 512, 32, 20000, 0, "gauss"), (4, 8, 256, 16, 20000, 1, "gauss"), (2, 4, 256, 128, 5000, 0, "rosen_mix"),
         (2, 16, 1024, 32, 10000, 0, "gauss"), (4, 8, 256, 16, 10000, 0, "rosen_mix"),
         (2, 8, 512, 64, 10000, 0, "gauss"), (2, 8, 256, 32, 10000, 1, "gauss_periodic"), (2, 8, 256, 64, 5000, 0, "gauss_periodic"))
if len(sys.argv) > 1:
    import importlib
    nr, T, W, D, iters, delay = (int(v) for v in sys.argv[1:7]); model = sys.argv[7]
    os.environ["PIPE_TEST_DELAY"] = str(delay); os.environ["PIPE_TEST_MODEL"] = model
    pw = importlib.import_module("pipeline_worker")
    from eryn_amd.ladder import LadderPipeline, rung_partition
    e = pw.make(T, W, D)
    if delay: LadderPipeline.connect_local([e])
    done = 0
    while done < iters:
        k = min(777, iters - done); e.step(k); done += k
    ref = pw.snapshot(e); e.close()
    _, bounds = rung_partition(T, nr)
    engs = [pw.make(T, W, D, b) for b in bounds]
    LadderPipeline.connect_local(engs)
    done = 0
    while done < iters:
        k = min(40, iters - done)
        for g in engs: g.step(k)
        for g in engs: g.synchronize()
        done += k
    snaps = [pw.snapshot(g) for g in engs]
    out = {k: np.concatenate([s[k] for s in snaps], axis=0) for k in ("x", "L", "P", "accepted")}
    for k in ("betas", "swaps_total", "swaps_last"): out[k] = snaps[0][k]
    bad = [k for k in KEYS if not np.array_equal(ref[k], out[k])]
    print(f"{nr} ranks, {T}x{W}x{D}, {iters} iterations, delay {delay}, {model}: pipeline == one context: {not bad} {bad}", flush=True)
else:
    import subprocess
    for c in cases:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(v) for v in c], capture_output=True, text=True, timeout=900)
        print((r.stdout.strip().splitlines() or ["(no output)"])[-1], r.stderr.strip()[-300:] if r.returncode else "", flush=True)
