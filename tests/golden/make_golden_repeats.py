"""Build-container only: fixture for the sampler-loop contract beyond one proposal per stored step - the REAL reference
(/root/reference/src, imported, never copied) run with ``num_repeats_in_model = 3`` and ``thin_by = 2`` (ensemble.py:243-256,
963-1045): every stored state, the backend's accept / swap accumulators and the move's own counters.

    python tests/golden/make_golden_repeats.py        ->  tests/golden/r1_repeats3_thin2.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for m in ("corner", "seaborn"):                 # imported unconditionally by eryn/utils/plot.py
    sys.modules.setdefault(m, types.ModuleType(m))
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True

from eryn.ensemble import EnsembleSampler  # noqa: E402
from eryn.prior import ProbDistContainer, uniform_dist  # noqa: E402


def log_like_vec(x, mu, invcov):            # tests/test_eryn.py:33-35, batched
    diff = x - mu
    return -0.5 * (diff * np.dot(invcov, diff.T).T).sum(axis=1)


def capture(name, T, W, D, nsteps, repeats, thin_by, box=50.0, seed_construct=123, seed_run=456):
    rs = np.random.RandomState(0)
    A = rs.randn(D, D)
    mu = 0.1 * rs.randn(D)
    invcov = np.linalg.inv(A @ A.T / D + np.eye(D))
    np.random.seed(seed_construct)
    priors = ProbDistContainer({i: uniform_dist(-box, box) for i in range(D)})
    s = EnsembleSampler(W, D, log_like_vec, priors, args=[mu, invcov], vectorize=True,
                        tempering_kwargs=dict(ntemps=T), num_repeats_in_model=repeats)
    x0 = np.random.RandomState(1).randn(T, W, D)
    out = dict(T=T, W=W, D=D, nsteps=nsteps, repeats=repeats, thin_by=thin_by, box=float(box), seed_construct=seed_construct,
               seed_run=seed_run, mu=mu, invcov=invcov, x0=x0)
    np.random.seed(seed_run)
    it = 0
    for state in s.sample(x0, iterations=nsteps, thin_by=thin_by, store=True):
        pre = f"it{it}_"
        out[pre + "x"] = state.branches["model_0"].coords[:, :, 0, :].copy()
        out[pre + "L"], out[pre + "P"] = state.log_like.copy(), state.log_prior.copy()
        out[pre + "betas"] = state.betas.copy()
        out[pre + "swaps_accepted"] = np.array(s.temperature_control.swaps_accepted, copy=True)
        out[pre + "backend_accepted"] = np.array(s.backend.accepted, copy=True)
        out[pre + "backend_swaps"] = np.array(s.backend.swaps_accepted, copy=True)
        it += 1
    out["move_accepted"] = np.array(s.moves[0].accepted, copy=True)
    out["num_proposals"] = int(s.moves[0].num_proposals)
    out["chain"] = s.get_chain()["model_0"][:, :, :, 0, :].copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path) // 1024, "KiB; proposals", out["num_proposals"], "mean accept",
          out["move_accepted"].mean() / out["num_proposals"])


if __name__ == "__main__":
    capture("r1_repeats3_thin2", T=3, W=16, D=4, nsteps=6, repeats=3, thin_by=2)
