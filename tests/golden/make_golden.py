#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (the reference tree does not exist on the GPU
box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports mikekatz04/Eryn from /root/reference/src (read-only), drives
``EnsembleSampler.sample`` with the default ``StretchMove`` (+ tempering), and
records, per iteration, every random draw of both streams (R = the sampler's
RandomState, G = the global np.random) and every intermediate of the hot path.
The fixtures are data only: inputs and expected outputs.  They pin
``oracle/eryn_oracle.py`` (tests/test_oracle_golden.py), which in turn is the
checker for the HIP path.

Capture points (all file:line under /root/reference/src/eryn):
  R draws           proxy around sampler._random        (ensemble.py:971, stretch.py:93-99,129-132, red_blue.py:294)
  G draws           wrappers on np.random.shuffle/permutation/uniform (red_blue.py:124, tempering.py:526-535)
  q, factors        StretchMove.get_proposal return     (stretch.py:160-231)
  logp, logl        sampler.compute_log_prior/_like     (ensemble.py:1127,1219)
  keep              Move.update(accepted, subset)       (move.py:472-703)
  state pre-PT      TemperatureControl.temper_comps in  (tempering.py:598)
  sel (as indices)  TemperatureControl.do_swaps_indexing(tempering.py:351)
  state, betas      yielded State per iteration         (ensemble.py:1045)
"""
import os
import sys
import types

for _m in ("corner", "seaborn"):           # imported unconditionally by eryn/utils/plot.py
    sys.modules[_m] = types.ModuleType(_m)
sys.path.insert(0, "/root/reference/src")

import numpy as np                          # noqa: E402
from eryn.ensemble import EnsembleSampler   # noqa: E402
from eryn.prior import ProbDistContainer, uniform_dist  # noqa: E402
from eryn.moves.tempering import make_ladder  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def gaussian_problem(D, dense=True):
    """SURVEY 8d synthetic Gaussian: mu = 0.1 randn, Sigma = A A^T / D + I."""
    rs = np.random.RandomState(0)
    A = rs.randn(D, D)
    mu = 0.1 * rs.randn(D)
    if dense:
        cov = A @ A.T / D + np.eye(D)
    else:
        cov = np.eye(D)
    return mu, np.linalg.inv(cov)


def log_like_vec(x, mu, invcov):            # tests/test_eryn.py:33-35, batched
    diff = x - mu
    return -0.5 * (diff * np.dot(invcov, diff.T).T).sum(axis=1)


def log_like_single(x, mu, invcov):         # tests/test_eryn.py:33-35 verbatim semantics
    diff = x - mu
    return -0.5 * (diff * np.dot(invcov, diff.T).T).sum()


class RProxy:
    """Recording proxy around the sampler-owned RandomState."""

    def __init__(self, rs, log):
        self._rs, self._log = rs, log

    def __getattr__(self, name):
        return getattr(self._rs, name)

    def choice(self, *a, **k):
        out = self._rs.choice(*a, **k)
        self._log.append(("choice", None))
        return out

    def randint(self, *a, **k):
        out = self._rs.randint(*a, **k)
        self._log.append(("randint", np.array(out, copy=True)))
        return out

    def rand(self, *a, **k):
        out = self._rs.rand(*a, **k)
        self._log.append(("rand", np.array(out, copy=True)))
        return out


def capture(name, T, W, D, nsteps, box, dense=True, vectorize=True, seed_construct=123,
            seed_run=456, tempering_kwargs=None, x0_scale=1.0, x0_uniform=False, keep_q=True, nsplits=2):
    mu, invcov = gaussian_problem(D, dense=dense)
    np.random.seed(seed_construct)          # R := snapshot of G at construction (ensemble.py:604,651-652)
    priors = ProbDistContainer({i: uniform_dist(-box, box) for i in range(D)})
    kw = {}
    tempered = T > 1 or tempering_kwargs is not None
    if tempered:
        tk = dict(ntemps=T)
        tk.update(tempering_kwargs or {})
        kw["tempering_kwargs"] = tk
    if nsplits != 2:                        # RedBlueMove(nsplits=...): more than two sets (red_blue.py:41-47,148)
        from eryn.moves import StretchMove
        kw["moves"] = StretchMove(nsplits=nsplits)
    s = EnsembleSampler(W, D, log_like_vec if vectorize else log_like_single, priors,
                        args=[mu, invcov], vectorize=vectorize, **kw)
    if x0_uniform:
        x0 = np.random.RandomState(1).uniform(-x0_scale * box, x0_scale * box, size=(T, W, D))
    else:
        x0 = x0_scale * np.random.RandomState(1).randn(T, W, D)

    rlog, glog, plog = [], [], []
    s._random = RProxy(s._random, rlog)

    orig_shuffle, orig_perm, orig_unif = np.random.shuffle, np.random.permutation, np.random.uniform

    def shuffle(x):
        orig_shuffle(x)
        glog.append(("shuffle", np.array(x, copy=True)))

    def permutation(n):
        out = orig_perm(n)
        glog.append(("permutation", np.array(out, copy=True)))
        return out

    def uniform(*a, **k):
        out = orig_unif(*a, **k)
        glog.append(("uniform", np.array(out, copy=True)))
        return out

    move = s.moves[0]
    orig_get_proposal, orig_update = move.get_proposal, move.update
    orig_lp, orig_ll = s.compute_log_prior, s.compute_log_like

    def get_proposal(s_all, c_all, random, **k):
        q, factors = orig_get_proposal(s_all, c_all, random, **k)
        plog.append(("q", q["model_0"][:, :, 0, :].copy()))
        plog.append(("factors", factors.copy()))
        return q, factors

    def compute_log_prior(coords, **k):
        out = orig_lp(coords, **k)
        plog.append(("logp", out.copy()))
        return out

    def compute_log_like(coords, **k):
        out = orig_ll(coords, **k)
        plog.append(("logl", out[0].copy()))
        return out

    def update(old_state, new_state, accepted, subset=None):
        plog.append(("keep", np.take_along_axis(accepted, subset, axis=1).copy()))
        plog.append(("subset", subset.copy()))
        out = orig_update(old_state, new_state, accepted, subset=subset)
        plog.append(("x_after_split", out.branches["model_0"].coords[:, :, 0, :].copy()))
        return out

    move.get_proposal, move.update = get_proposal, update
    s.compute_log_prior, s.compute_log_like = compute_log_prior, compute_log_like

    tc = s.temperature_control
    if tc is not None:
        orig_tc, orig_dsi = tc.temper_comps, tc.do_swaps_indexing

        def temper_comps(state, **k):
            plog.append(("pre_pt", (state.branches["model_0"].coords[:, :, 0, :].copy(),
                                    state.log_like.copy(), state.log_prior.copy())))
            return orig_tc(state, **k)

        def do_swaps_indexing(i, iperm_sel, i1perm_sel, *a, **k):
            plog.append(("swap_idx", (i, iperm_sel.copy(), i1perm_sel.copy())))
            return orig_dsi(i, iperm_sel, i1perm_sel, *a, **k)

        tc.temper_comps, tc.do_swaps_indexing = temper_comps, do_swaps_indexing

    out = dict(T=T, W=W, D=D, nsteps=nsteps, box=float(box), dense=dense, vectorize=vectorize, nsplits=nsplits,
               seed_construct=seed_construct, seed_run=seed_run, x0_scale=float(x0_scale),
               x0_uniform=bool(x0_uniform),
               mu=mu, invcov=invcov, x0=x0, a=float(move.a))
    if tc is not None:
        out["betas0"] = np.array(tc.betas, copy=True)
        out["adaptive"], out["permute"] = bool(tc.adaptive), bool(tc.permute)

    np.random.seed(seed_run)                # G for the run
    np.random.shuffle, np.random.permutation, np.random.uniform = shuffle, permutation, uniform
    try:
        it = 0
        # the initial log_prior/log_like evaluation goes through the wrapped fns too
        for state in s.sample(x0, iterations=nsteps, store=False):
            pre = f"it{it}_"
            # ---- R log: choice, then per split randint, rand(zz), rand(acc)
            assert [k for k, _ in rlog] == ["choice"] + ["randint", "rand", "rand"] * nsplits, rlog
            for sp in range(nsplits):
                out[pre + f"rint{sp}"] = rlog[1 + 3 * sp][1]
                out[pre + f"u_zz{sp}"] = rlog[2 + 3 * sp][1]
                out[pre + f"u_acc{sp}"] = rlog[3 + 3 * sp][1]
            rlog.clear()
            # ---- G log: T shuffles, then (perm, perm, uniform) x (T-1)
            kinds = [k for k, _ in glog]
            nperm = (2 if (tc is not None and tc.permute) else 0)
            expect = ["shuffle"] * T
            if tc is not None:
                expect += (["permutation"] * nperm + ["uniform"]) * (T - 1)
            assert kinds == expect, (kinds, expect)
            out[pre + "labels"] = np.stack([v for k, v in glog[:T]])
            if tc is not None and T > 1:
                rest = glog[T:]
                step = nperm + 1
                if nperm:
                    out[pre + "iperm"] = np.stack([rest[j * step][1] for j in range(T - 1)])
                    out[pre + "i1perm"] = np.stack([rest[j * step + 1][1] for j in range(T - 1)])
                out[pre + "u_swap"] = np.stack([rest[j * step + nperm][1] for j in range(T - 1)])
            glog.clear()
            # ---- path log
            pl = list(plog)
            plog.clear()
            if it == 0:
                # initial evaluation: logp then logl of x0 (ensemble.py:898-912)
                assert pl[0][0] == "logp" and pl[1][0] == "logl"
                out["P0"], out["L0"] = pl[0][1], pl[1][1]
                pl = pl[2:]
            names = [k for k, _ in pl]
            per_split = ["q", "factors", "logp", "logl", "keep", "subset", "x_after_split"]
            expect = per_split * nsplits + (["pre_pt"] if tc is not None else [])
            assert names[:len(expect)] == expect, names
            for sp in range(nsplits):
                blk = dict(pl[7 * sp:7 * sp + 7])
                if keep_q:
                    out[pre + f"q{sp}"] = blk["q"]
                out[pre + f"factors{sp}"] = blk["factors"]
                out[pre + f"logp{sp}"] = blk["logp"]
                out[pre + f"logl{sp}"] = blk["logl"]
                out[pre + f"keep{sp}"] = blk["keep"]
                out[pre + f"S{sp}"] = blk["subset"]
            if tc is not None:
                xs, Ls, Ps = pl[7 * nsplits][1]
                out[pre + "L_stretch"], out[pre + "P_stretch"] = Ls, Ps
                if keep_q:
                    out[pre + "x_stretch"] = xs
                swaps = pl[7 * nsplits + 1:]
                assert all(k == "swap_idx" for k, _ in swaps) and len(swaps) == T - 1
                sel = np.zeros((T - 1, W), dtype=bool)
                for j, (_, (i, a_, b_)) in enumerate(swaps):
                    assert i == T - 1 - j
                    ip = out[pre + "iperm"][j] if nperm else np.arange(W)
                    i1p = out[pre + "i1perm"][j] if nperm else np.arange(W)
                    m = np.isin(ip, a_)
                    assert np.array_equal(ip[m], a_) and np.array_equal(i1p[m], b_)
                    sel[j] = m
                out[pre + "sel"] = sel
                out[pre + "swaps_accepted"] = np.array(tc.swaps_accepted, copy=True)
                out[pre + "betas"] = np.array(tc.betas, copy=True)
            out[pre + "x"] = state.branches["model_0"].coords[:, :, 0, :].copy()
            out[pre + "L"] = state.log_like.copy()
            out[pre + "P"] = state.log_prior.copy()
            it += 1
        out["accepted_total"] = np.array(move.accepted, copy=True)
        out["num_proposals"] = int(move.num_proposals)
    finally:
        np.random.shuffle, np.random.permutation, np.random.uniform = orig_shuffle, orig_perm, orig_unif
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, "
          f"mean accept {out['accepted_total'].mean() / nsteps:.3f}")


def ladders():
    out = {}
    for D in (1, 5, 32, 64, 100, 101, 128):
        for T in (2, 4, 16, 32, 64):
            out[f"D{D}_T{T}"] = make_ladder(D, ntemps=T)
    out["D6_T5_inf"] = make_ladder(6, ntemps=5, Tmax=np.inf)
    out["D6_Tmax50"] = make_ladder(6, Tmax=50.0)
    out["D32_T8_Tmax1000"] = make_ladder(32, ntemps=8, Tmax=1000.0)
    np.savez_compressed(os.path.join(HERE, "ladders.npz"), **out)
    print("ladders:", len(out))


if __name__ == "__main__":
    ladders()
    # F1 plumbing = BASELINE config 1 / tests/test_eryn.py::test_base shape (no tempering, non-vectorised)
    capture("f1_plumbing", T=1, W=32, D=5, nsteps=20, box=5.0, dense=False, vectorize=False)
    # F2 PT, dense covariance
    capture("f2_pt", T=4, W=20, D=6, nsteps=25, box=50.0)
    # F3 odd walker count -> uneven halves (red_blue.py:121-124)
    capture("f3_oddW", T=3, W=33, D=4, nsteps=15, box=50.0)
    # F4 narrow box: many -inf-prior proposals -> -1e300 fill (ensemble.py:1486-1513)
    capture("f4_narrowbox", T=5, W=21, D=6, nsteps=20, box=1.0, x0_scale=0.95, x0_uniform=True)
    # F5 adaptive=False / permute=False variants (tempering.py:525-532,632)
    capture("f5_noadapt", T=3, W=16, D=4, nsteps=10, box=50.0, tempering_kwargs=dict(adaptive=False))
    capture("f5_nopermute", T=3, W=16, D=4, nsteps=10, box=50.0, tempering_kwargs=dict(permute=False))
    # F6 medium teacher-forced steps at the kernel's native row width (D = 32)
    capture("f6_medium", T=8, W=128, D=32, nsteps=2, box=50.0)
    # F7 Tmax=inf ladder (beta = 0 rung: 0 * -1e300 and friends)
    capture("f7_tmaxinf", T=4, W=24, D=5, nsteps=12, box=2.0, x0_scale=0.9, x0_uniform=True,
            tempering_kwargs=dict(Tmax=np.inf))
    # F8 three sets (RedBlueMove(nsplits=3)): uneven sets 6 / 6 / 5, the complement list = the other sets in set order
    capture("f8_nsplits3", T=3, W=17, D=4, nsteps=12, box=50.0, nsplits=3)
