"""Where a short hens_step call's time goes (driver shape: blocks of 20 iterations between synchronisations)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from tools.quick_bench import problem, ladder
import os; os.environ.pop("HENS_STEP_EVENTS", None)      # (quick_bench sets it on import: an event pair per call, and the HIP stream)
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood

T, W, D = 16, 4096, 32
mu, invcov, cov = problem(D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T))
eng.eval_state(); eng.step(200); eng.synchronize()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def run(name, pre, post, n=40):
    ts = []
    for _ in range(n):
        pre()
        t0 = time.perf_counter()
        eng.step(K)
        post()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"{name:58s} median {np.median(ts):7.1f} us per block = {np.median(ts)/K:6.2f} us/iter   (min {ts.min():.1f})")

ts = lambda: torch.cuda.synchronize()
run("bench.py as is: sync x2 | step, eng.sync, torch.sync x2", lambda: (ts(), ts()), lambda: (eng.synchronize(), ts(), ts()))
run("sync | step, torch.sync", ts, ts)
run("sync | step, eng.sync", ts, eng.synchronize)
t0 = time.perf_counter(); eng.step(K); t1 = time.perf_counter(); eng.synchronize()
print(f"host time of one hens_step({K}) call: {(t1 - t0)*1e6:.1f} us")
t0 = time.perf_counter(); eng.step(2000); eng.synchronize(); print(f"long call: {(time.perf_counter()-t0)/2000*1e6:.2f} us/iter")
