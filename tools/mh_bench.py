"""Exploratory timing of the Gaussian MH move and the stretch+MH mix in hens_step (cfg-2 shape)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from tools.quick_bench import problem, ladder
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood

T, W, D = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 4096, 32)))
mu, invcov, cov = problem(D)
for kind, scale, weight in (("iso", 0.3, 1.0), ("diag", np.full(D, 0.3), 1.0), ("full", np.linalg.cholesky(0.05 * cov), 1.0),
                            ("iso", 0.3, 0.5), (None, None, 0.0)):
    eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
    eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T))
    eng.eval_state()
    if kind:
        eng.set_mh_proposal(kind, scale, weight)
    eng.step(200)
    eng.synchronize()
    t0 = time.perf_counter()
    n = 2000
    eng.step(n)
    eng.synchronize()
    dt = time.perf_counter() - t0
    acc = eng.mh_counters() if kind else None
    frac = float(acc["accepted"].mean() / max(acc["num_proposals"], 1)) if kind else float("nan")
    print(f"move mix {str(kind):5s} weight {weight:3.1f}: {dt / n * 1e6:7.2f} us/iter  {T * W * n / dt:.3e} walker-steps/s   MH acceptance {frac:.3f}")
    eng.close()
