"""Exploratory timing of hens_step on one GPU (not the contract bench; see bench.py)."""
import argparse
import os
import time

os.environ.setdefault("HENS_STEP_EVENTS", "1")     # the "device" figure below: an event pair around the hens_step call

import numpy as np

from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood, RosenbrockLikelihood


def problem(D):
    rs = np.random.RandomState(0)
    A = rs.randn(D, D)
    mu = 0.1 * rs.randn(D)
    cov = A @ A.T / D + np.eye(D)
    return mu, np.linalg.inv(cov), cov


def ladder(D, T):
    from eryn_amd.moves.tempering import make_ladder
    return make_ladder(D, ntemps=T)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=16)
    ap.add_argument("--W", type=int, default=4096)
    ap.add_argument("--D", type=int, default=32)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--prof", type=int, default=0)
    ap.add_argument("--stats", type=int, default=0)
    ap.add_argument("--like", default="dense")
    ap.add_argument("--adaptive", type=int, default=1)
    a = ap.parse_args()
    T, W, D = a.T, a.W, a.D
    mu, invcov, cov = problem(D)
    like = RosenbrockLikelihood(D) if a.like == 'rosen' else GaussianLikelihood(mu, invcov if a.like == 'dense' else np.diag(invcov).copy())
    eng = HipEnsemble(T, W, D, like, -50.0 if a.like != 'rosen' else -5.0, 50.0 if a.like != 'rosen' else 5.0, seed=2024,
                      adaptive=bool(a.adaptive))
    x0 = np.random.RandomState(1).randn(T, W, D)
    eng.upload(x0, betas=ladder(D, T))
    eng.eval_state()
    eng.step(a.warmup)
    eng.synchronize()
    eng.set_profiling(bool(a.prof))
    t0 = time.perf_counter()
    eng.step(a.steps)
    eng.synchronize()
    dt = time.perf_counter() - t0
    tm = eng.timing()
    print(f"T={T} W={W} D={D}: {a.steps} iters in {dt*1e3:.1f} ms host, {tm['total_ms']:.1f} ms device "
          f"-> {T*W*a.steps/dt:.3e} walker-steps/s, {dt/a.steps*1e6:.1f} us/iter")
    if a.prof:
        print({k: v for k, v in tm.items()})
        print(f"stretch avg {tm['stretch_ms']/max(tm['n_stretch'],1)*1e3:.2f} us, pt avg "
              f"{tm['pt_ms']/max(tm['n_pt'],1)*1e3:.2f} us, plan avg {tm['plan_ms']/max(tm['n_plan'],1)*1e3:.1f} us")
    c = eng.counters()
    print("acceptance per rung:", np.round(c["accepted"].mean(axis=1) / c["num_proposals"], 3))
    print("swap fraction per pair:", np.round(c["swaps_total"] / W / max(c["adapt_time"], 1), 3))
    if a.stats:
        x, L, P, betas = eng.download()
        print("betas:", np.round(betas, 4))
        xc = x[0]
        print("cold mean err max:", np.abs(xc.mean(0) - mu).max(), " expected ~", np.sqrt(np.diag(cov).max() / W) * 3)
        print("cold cov rel err (fro):", np.linalg.norm(np.cov(xc.T) - cov) / np.linalg.norm(cov))
        print("mean logl cold:", L[0].mean(), " expected ~", -D / 2)
