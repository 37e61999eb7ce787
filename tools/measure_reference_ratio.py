"""Build-container only: time the REAL reference (/root/reference/src, imported, never copied) against the repo's
NumPy restatement (oracle/eryn_oracle.py) on identical inputs, and write profiles/cpu_reference_ratio.json.
bench.py's cpu_baseline reads that file as a static, labelled figure: the reference cannot travel to the GPU box."""
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for m in ("corner", "seaborn"):                 # imported unconditionally by eryn/utils/plot.py
    sys.modules.setdefault(m, types.ModuleType(m))
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True

from eryn.ensemble import EnsembleSampler  # noqa: E402
from eryn.prior import ProbDistContainer, uniform_dist  # noqa: E402

from bench import gaussian_problem  # noqa: E402
from oracle import eryn_oracle as orc  # noqa: E402


def time_reference(T, W, D, iters):
    mu, invcov = gaussian_problem(D)

    def loglike(x, mu, invcov):
        diff = x - mu
        return -0.5 * (diff * np.dot(invcov, diff.T).T).sum(axis=1)

    np.random.seed(123)
    kw = dict(tempering_kwargs=dict(ntemps=T)) if T > 1 else {}
    s = EnsembleSampler(W, D, loglike, ProbDistContainer({i: uniform_dist(-50, 50) for i in range(D)}),
                        args=[mu, invcov], vectorize=True, **kw)
    np.random.seed(456)
    x0 = np.random.RandomState(1).randn(T, W, D)
    st = s.run_mcmc(x0 if T > 1 else x0[0], 1, store=False)
    t0 = time.perf_counter()
    s.run_mcmc(st, iters, store=False)
    return (time.perf_counter() - t0) / iters


def time_oracle(T, W, D, iters):
    mu, invcov = gaussian_problem(D)
    o = orc.OracleSampler(np.random.RandomState(1).randn(T, W, D), lambda x: orc.gaussian_log_like(x, mu, invcov),
                          np.full(D, -50.0), np.full(D, 50.0), np.random.RandomState(123), np.random.RandomState(456),
                          betas=orc.make_ladder(D, ntemps=T) if T > 1 else None)
    o.iteration()
    t0 = time.perf_counter()
    for _ in range(iters):
        o.iteration()
    return (time.perf_counter() - t0) / iters


if __name__ == "__main__":
    # config 2 (the headline), config 1 (the reference's own CPU-runnable case) and one GPU's shard of config 3
    shapes = {}
    for T, W, D, iters in ((16, 4096, 32, 12), (1, 32, 5, 400), (8, 16384, 64, 3)):
        tr, to = time_reference(T, W, D, iters), time_oracle(T, W, D, iters)
        shapes[f"{T}x{W}x{D}"] = {"iterations": iters, "reference_ms_per_iter": tr * 1e3, "oracle_ms_per_iter": to * 1e3,
                                  "reference_over_port": tr / to}
        print(T, W, D, shapes[f"{T}x{W}x{D}"], flush=True)
    c2 = shapes["16x4096x32"]
    out = {"config": {"ntemps": 16, "nwalkers": 4096, "ndim": 32, "iterations": c2["iterations"]},
           "reference_ms_per_iter": c2["reference_ms_per_iter"], "oracle_ms_per_iter": c2["oracle_ms_per_iter"],
           "reference_over_port": c2["reference_over_port"], "shapes": shapes,
           "host": f"build container, os.cpu_count()={os.cpu_count()}, numpy {np.__version__}",
           "note": "time ratio on identical inputs; the honest Eryn-CPU figure on another host is the oracle's rate divided by this"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "cpu_reference_ratio.json"), "w"), indent=1)
