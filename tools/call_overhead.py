"""Per-call cost of hens_step on the AQL queue: median wall time of step(K) + synchronize for several K, fit a + b K.
  python tools/call_overhead.py [T W D]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from tools.quick_bench import problem, ladder
os.environ.pop("HENS_STEP_EVENTS", None)
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood
T, W, D = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 4096, 32)))
mu, invcov, cov = problem(D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T))
eng.eval_state(); eng.step(500); eng.synchronize()
Ks, ts = [1, 2, 5, 10, 20, 50, 200], []
for K in Ks:
    reps = max(30, 2000 // K)
    v = []
    for _ in range(reps):
        t0 = time.perf_counter(); eng.step(K); eng.synchronize(); v.append(time.perf_counter() - t0)
    ts.append(np.median(v) * 1e6)
b, a = np.polyfit(Ks, ts, 1)
print(" ".join(f"K={k}: {t:.1f}" for k, t in zip(Ks, ts)), f"| fit {b:.3f} us/iteration + {a:.1f} us/call")
eng.close()
