#!/usr/bin/env python
"""Golden fixtures for the Metropolis-Hastings moves from the REAL reference.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mh.py

Drives the reference's ``EnsembleSampler`` (imported read-only from /root/reference/src) with
``GaussianMove`` (isotropic / diagonal / full covariance; vector / random / sequential mode; factor)
alone and mixed with ``StretchMove`` by weight, and records per iteration
  * which move ``self._random.choice(self._moves, p=self._weights)`` picked     (ensemble.py:971)
  * every draw of the sampler's RandomState R, in order, with its value         (gaussian.py:197-270,
    mh.py:157, stretch.py:93-132, red_blue.py:294)
  * every draw of the global np.random stream G                                 (red_blue.py:124, tempering.py:526-535)
  * proposal q, log-prior, log-like, accept mask of the MH proposals            (mh.py:108-157)
  * the yielded state (x, log_like, log_prior), betas, swaps_accepted            (ensemble.py:1045)
The files are data only (inputs + expected outputs); tests/test_oracle_golden.py pins the oracle's MH
restatement on them, tests/test_hip_mh.py the HIP path.
"""
import os
import sys
import types

for _m in ("corner", "seaborn"):
    sys.modules[_m] = types.ModuleType(_m)
sys.path.insert(0, "/root/reference/src")

import numpy as np                                   # noqa: E402
from eryn.ensemble import EnsembleSampler            # noqa: E402
from eryn.moves import GaussianMove, StretchMove     # noqa: E402
from eryn.prior import ProbDistContainer, uniform_dist  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import gaussian_problem, log_like_vec  # noqa: E402


class RLog:
    """Proxy that records (method, result) of every call on the sampler-owned RandomState."""

    def __init__(self, rs, log):
        self._rs, self._log = rs, log

    def __getattr__(self, name):
        attr = getattr(self._rs, name)
        if not callable(attr) or name in ("get_state", "set_state"):
            return attr

        def wrapped(*a, **k):
            out = attr(*a, **k)
            self._log.append((name, None if name == "choice" else np.array(out, copy=True)))
            return out
        return wrapped


def capture(name, T, W, D, nsteps, moves_spec, box=6.0, seed_construct=321, seed_run=654, tempering_kwargs=None,
            periodic=None):
    """moves_spec: list of ("stretch", weight) | ("gauss", weight, dict(cov=..., mode=..., factor=...)).
    periodic: {parameter index: period} of the single branch (ensemble.py:165-168) or None."""
    mu, invcov = gaussian_problem(D)
    np.random.seed(seed_construct)
    priors = ProbDistContainer({i: uniform_dist(-box, box) for i in range(D)})
    moves = []
    for spec in moves_spec:
        if spec[0] == "stretch":
            moves.append((StretchMove(a=2.0), spec[1]))
        else:
            kw = dict(spec[2])
            cov = kw.pop("cov")
            moves.append((GaussianMove({"model_0": cov}, **kw), spec[1]))
    kw = {}
    if T > 1 or tempering_kwargs is not None:
        tk = dict(ntemps=T)
        tk.update(tempering_kwargs or {})
        kw["tempering_kwargs"] = tk
    if periodic is not None:
        kw["periodic"] = {"model_0": dict(periodic)}
    s = EnsembleSampler(W, D, log_like_vec, priors, args=[mu, invcov], vectorize=True, moves=moves, **kw)
    h = min(2.0, 0.9 * box)
    x0 = np.random.RandomState(1).uniform(-h, h, size=(T, W, D))
    rlog, glog, plog = [], [], []
    s._random = RLog(s._random, rlog)
    orig = (np.random.shuffle, np.random.permutation, np.random.uniform)

    def shuffle(x):
        orig[0](x)
        glog.append(("shuffle", np.array(x, copy=True)))

    def permutation(n):
        out = orig[1](n)
        glog.append(("permutation", np.array(out, copy=True)))
        return out

    def uniform(*a, **k):
        out = orig[2](*a, **k)
        glog.append(("uniform", np.array(out, copy=True)))
        return out

    chosen = []
    for idx, m in enumerate(s.moves):
        def make(idx, m):
            orig_propose = m.propose

            def propose(model, state):
                chosen.append(idx)
                return orig_propose(model, state)
            return propose
        m.propose = make(idx, m)
        if isinstance(m, GaussianMove):
            def make_gp(m):
                orig_gp = m.get_proposal

                def get_proposal(coords, random, **k):
                    q, f = orig_gp(coords, random, **k)
                    plog.append(("mh_q", q["model_0"][:, :, 0, :].copy()))
                    return q, f
                return get_proposal
            m.get_proposal = make_gp(m)
            def make_up(m):
                orig_up = m.update

                def update(old_state, new_state, accepted, subset=None):
                    plog.append(("mh_logl", new_state.log_like.copy()))
                    plog.append(("mh_logp", new_state.log_prior.copy()))
                    plog.append(("mh_keep", accepted.copy()))
                    return orig_up(old_state, new_state, accepted, subset=subset)
                return update
            m.update = make_up(m)
    tc = s.temperature_control
    out = dict(T=T, W=W, D=D, nsteps=nsteps, box=float(box), seed_construct=seed_construct, seed_run=seed_run,
               mu=mu, invcov=invcov, x0=x0, weights=np.array(s.weights, copy=True), nmoves=len(s.moves))
    for i, spec in enumerate(moves_spec):
        out[f"move{i}_kind"] = spec[0]
        if spec[0] == "gauss":
            out[f"move{i}_cov"] = np.asarray(spec[2]["cov"], dtype=np.float64)
            out[f"move{i}_mode"] = spec[2].get("mode", "vector")
            out[f"move{i}_factor"] = float(spec[2]["factor"]) if spec[2].get("factor") is not None else np.nan
    if tc is not None:
        out["betas0"] = np.array(tc.betas, copy=True)
    if periodic is not None:
        out["period"] = np.array([float(periodic.get(d, 0.0)) for d in range(D)])
    np.random.seed(seed_run)
    np.random.shuffle, np.random.permutation, np.random.uniform = shuffle, permutation, uniform
    try:
        it = 0
        for state in s.sample(x0, iterations=nsteps, store=False):
            pre = f"it{it}_"
            out[pre + "move"] = chosen[-1]
            assert rlog[0][0] == "choice"
            out[pre + "r_kinds"] = np.array([k for k, _ in rlog[1:]])
            for j, (_, v) in enumerate(rlog[1:]):
                out[pre + f"r{j}"] = v
            rlog.clear()
            out[pre + "g_kinds"] = np.array([k for k, _ in glog])
            for j, (_, v) in enumerate(glog):
                out[pre + f"g{j}"] = v
            glog.clear()
            for k, v in plog:
                out[pre + k] = v
            plog.clear()
            out[pre + "x"] = state.branches["model_0"].coords[:, :, 0, :].copy()
            out[pre + "L"] = state.log_like.copy()
            out[pre + "P"] = state.log_prior.copy()
            if tc is not None:
                out[pre + "betas"] = np.array(tc.betas, copy=True)
                out[pre + "swaps_accepted"] = np.array(tc.swaps_accepted, copy=True)
            it += 1
        for i, m in enumerate(s.moves):
            out[f"move{i}_accepted"] = np.array(m.accepted, copy=True)
            out[f"move{i}_num_proposals"] = int(m.num_proposals)
    finally:
        np.random.shuffle, np.random.permutation, np.random.uniform = orig
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    acc = [float(np.mean(out[f"move{i}_accepted"]) / max(out[f"move{i}_num_proposals"], 1)) for i in range(len(s.moves))]
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, moves picked {np.bincount(chosen, minlength=len(s.moves))}, accept {np.round(acc, 3)}")


if __name__ == "__main__":
    D = 4
    rs = np.random.RandomState(9)
    A = rs.randn(D, D)
    full = 0.05 * (A @ A.T / D + np.eye(D))
    capture("m1_gauss_iso", 3, 16, D, 6, [("gauss", 1.0, dict(cov=0.05))])
    # (a 1-D "diagonal" covariance cannot be constructed in the reference at this numpy: gaussian.py:144 calls
    #  np.linalg.cholesky on the 1-D scale and raises LinAlgError - so there is no diagonal fixture)
    capture("m3_gauss_full", 3, 16, D, 6, [("gauss", 1.0, dict(cov=full))])
    capture("m4_gauss_random_factor", 2, 12, D, 6, [("gauss", 1.0, dict(cov=0.2, mode="random", factor=2.0))])
    capture("m5_gauss_sequential", 2, 12, D, 6, [("gauss", 1.0, dict(cov=0.15, mode="sequential"))])
    capture("m6_mix", 4, 24, D, 12, [("stretch", 0.5), ("gauss", 0.5, dict(cov=0.05))])
    capture("m7_gauss_untempered", 1, 20, D, 5, [("gauss", 1.0, dict(cov=0.05))])
    capture("m8_mix_narrowbox", 3, 16, D, 8, [("stretch", 0.5), ("gauss", 0.5, dict(cov=0.5))], box=1.5, seed_run=655)
    # periodic parameters (ensemble.py:165-168, utils/periodic.py): distances the short way round in the stretch move
    # (stretch.py:136-141), every proposal wrapped into [0, period) (stretch.py:149-154, gaussian.py:110-115); the walkers
    # start on both sides of 0 and spread over more than half a period, so both branches of `distance` are taken
    capture("p1_stretch_periodic", 3, 16, D, 10, [("stretch", 1.0)], periodic={0: 2 * np.pi, 2: 1.5}, seed_run=656)
    capture("p2_mix_periodic", 4, 24, D, 12, [("stretch", 0.5), ("gauss", 0.5, dict(cov=0.3))],
            periodic={1: 2 * np.pi, 3: 3.0}, seed_run=657)
    capture("p3_stretch_periodic_untempered", 1, 20, 5, 8, [("stretch", 1.0)], periodic={4: 1.0}, seed_run=658)
