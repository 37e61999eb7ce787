#!/usr/bin/env python
"""Golden fixtures for the reversible-jump leaf-packing path from the REAL reference.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_rj.py

Drives the reference's ``EnsembleSampler`` (imported read-only from /root/reference/src) on the model of its own
``test_rj_multiple_branches`` (tests/test_eryn.py:341-507) in small: two branches (Gaussian pulses + sine waves) with a
variable number of leaves, in-model ``GaussianMove`` on the packed active leaves, ``DistributionGenerateRJ`` birth /
death per branch ("separate_branches"), tempering.  Recorded per iteration: the state after the in-model move and after
the RJ move (coordinates of every leaf slot, ``inds``, log-like, log-prior), both moves' accept masks, betas, swap counts.
The draws themselves are not stored: NumPy's legacy ``RandomState`` streams are version-stable, so the oracle
(oracle/eryn_oracle_rj.py) regenerates them from the two seeds and must land on the same states bit for bit - which also
pins the ORDER in which the path consumes the sampler's stream R and the global stream G.
The files are data only (inputs + expected outputs).
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
from eryn.prior import uniform_dist                  # noqa: E402
from eryn.state import State                         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


# the reference tests' own model functions (tests/test_eryn.py:38-92), imported - not restated - from the reference tree
sys.path.insert(0, "/root/reference/tests")
from test_eryn import combine_gaussians, combine_sine, log_like_fn_gauss_and_sine     # noqa: E402


GAUSS_BOX = [(2.5, 3.5), (-1.0, 1.0), (0.01, 0.21)]               # tests/test_eryn.py:432-443
SINE_BOX = [(0.5, 1.5), (1.0, 20.0), (0.0, 2 * np.pi)]


# Round 6: models whose branches have OTHER leaf widths than three (the reference's `ndims` per branch, ensemble.py:325-329) and a
# likelihood that is a plain Python function (oracle/eryn_oracle_rj.py) - the device library's hens_rj_set_model_general.
# name -> (branch names, boxes, starting leaves per branch, the likelihood's name in oracle/eryn_oracle_rj.py, the data's signal)
GENERAL_MODELS = {
    "ramp_burst": (["ramp", "burst"],
                   [[(-1.0, 1.0), (-2.0, 2.0)], [(0.5, 3.0), (-1.0, 1.0), (0.05, 0.5), (1.0, 8.0)]],
                   [np.array([[0.3, -0.8], [-0.2, 0.5], [0.1, 0.2]]),
                    np.array([[2.0, -0.3, 0.2, 3.0], [1.2, 0.4, 0.1, 6.0], [0.9, 0.0, 0.3, 2.0]])],
                   "ramp_burst_log_like",
                   lambda t: 0.3 - 0.8 * t + 2.0 * np.exp(-(((t + 0.3) / 0.2) ** 2)) * np.cos(2 * np.pi * 3.0 * (t + 0.3))),
    "offset": (["offset"], [[(-3.0, 3.0)]], [np.array([[0.7], [-0.4], [0.2], [1.1]])], "offset_log_like", lambda t: 0.3 + 0.0 * t),
}


def capture(name, T, W, nl_max, nl_min, nsteps, ndata=40, sigma=2.0, cov_factor=1e-4, n_init=(2, 1),
            seed_data=42, seed_construct=135, seed_run=246, rj_moves="separate_branches", in_model="gaussian", init_spread=1e-4,
            model="template"):
    """in_model: "gaussian" - GaussianMove on the packed leaves; "stretch" - the red / blue StretchMove over EVERY branch and leaf
    slot of a walker (stretch.py:160-231: one complement draw per branch, one zz per walker; red_blue.py:103-330).  rj_moves None:
    no reversible jump (the leaf masks stay as they start)."""
    if model in GENERAL_MODELS:
        return capture_general(name, T, W, nl_max, nl_min, nsteps, ndata, sigma, cov_factor, n_init, seed_data, seed_construct, seed_run,
                               rj_moves, in_model, init_spread, model)
    branch_names = ["gauss", "sine"]
    ndims = {"gauss": 3, "sine": 3}
    nleaves_max = dict(zip(branch_names, nl_max))
    nleaves_min = dict(zip(branch_names, nl_min))
    t = np.linspace(-1, 1, ndata)
    gauss_inj = np.array([[3.3, -0.2, 0.1], [2.6, -0.1, 0.1], [3.4, 0.0, 0.1], [2.9, 0.3, 0.1]])
    sine_inj = np.array([[1.3, 10.1, 1.0], [0.8, 4.6, 1.2]])
    rs = np.random.RandomState(seed_data)
    y = combine_gaussians(t, gauss_inj[:n_init[0]]) + combine_sine(t, sine_inj[:n_init[1]]) + sigma * rs.randn(ndata)
    coords = {k: np.zeros((T, W, nleaves_max[k], 3)) for k in branch_names}
    inds = {k: np.zeros((T, W, nleaves_max[k]), dtype=bool) for k in branch_names}
    for nn in range(n_init[0]):
        coords["gauss"][:, :, nn] = rs.multivariate_normal(gauss_inj[nn], np.diag(np.ones(3) * init_spread), size=(T, W))
        inds["gauss"][:, :, nn] = True
    for nn in range(n_init[1]):
        coords["sine"][:, :, nn] = rs.multivariate_normal(sine_inj[nn], np.diag(np.ones(3) * init_spread), size=(T, W))
        inds["sine"][:, :, nn] = True
    priors = {"gauss": {i: uniform_dist(*GAUSS_BOX[i]) for i in range(3)},
              "sine": {i: uniform_dist(*SINE_BOX[i]) for i in range(3)}}
    cov = {k: np.diag(np.ones(3)) * cov_factor for k in branch_names}
    boxes = None
    return run_capture(name, T, W, nl_max, nl_min, nsteps, ndata, sigma, cov_factor, seed_construct, seed_run, rj_moves, in_model, model,
                       branch_names, ndims, nleaves_max, nleaves_min, t, y, coords, inds, priors, cov, boxes)


def capture_general(name, T, W, nl_max, nl_min, nsteps, ndata, sigma, cov_factor, n_init, seed_data, seed_construct, seed_run, rj_moves,
                    in_model, init_spread, model):
    """The same capture for a GENERAL_MODELS entry: branches of 1 .. 4 parameters per leaf, the likelihood a Python function."""
    branch_names, boxes, inj, like_name, signal = GENERAL_MODELS[model]
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import oracle.eryn_oracle_rj as orj_
    like = getattr(orj_, like_name)
    ndims = {k: len(bx) for k, bx in zip(branch_names, boxes)}
    nleaves_max = dict(zip(branch_names, nl_max))
    nleaves_min = dict(zip(branch_names, nl_min))
    t = np.linspace(-1, 1, ndata)
    rs = np.random.RandomState(seed_data)
    y = signal(t) + sigma * rs.randn(ndata)
    coords = {k: np.zeros((T, W, nleaves_max[k], ndims[k])) for k in branch_names}
    inds = {k: np.zeros((T, W, nleaves_max[k]), dtype=bool) for k in branch_names}
    for bi, k in enumerate(branch_names):
        for nn in range(n_init[bi]):
            coords[k][:, :, nn] = rs.multivariate_normal(inj[bi][nn], np.diag(np.ones(ndims[k]) * init_spread), size=(T, W))
            inds[k][:, :, nn] = True
    priors = {k: {i: uniform_dist(*boxes[bi][i]) for i in range(ndims[k])} for bi, k in enumerate(branch_names)}
    cov = {k: np.diag(np.ones(ndims[k])) * cov_factor for k in branch_names}
    return run_capture(name, T, W, nl_max, nl_min, nsteps, ndata, sigma, cov_factor, seed_construct, seed_run, rj_moves, in_model, model,
                       branch_names, ndims, nleaves_max, nleaves_min, t, y, coords, inds, priors, cov, boxes, like=like)


def run_capture(name, T, W, nl_max, nl_min, nsteps, ndata, sigma, cov_factor, seed_construct, seed_run, rj_moves, in_model, model,
                branch_names, ndims, nleaves_max, nleaves_min, t, y, coords, inds, priors, cov, boxes, like=None):

    np.random.seed(seed_construct)          # R := snapshot of G at construction (ensemble.py:604,651-652)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")     # (ensemble.py:509-514: the reference advises against the stretch move under RJ - and runs it)
        # model "lorentz_chirp" (round 6): a likelihood the device library has no kernel for - oracle/eryn_oracle_rj.py's
        # lorentz_chirp_log_like, the function the -m gpu test hands to RJEnsembleSampler as the user's callable
        like = log_like_fn_gauss_and_sine if like is None else like
        if model == "lorentz_chirp":
            sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)) if os.path.isdir(os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle")) else "/root/repo")
            from oracle.eryn_oracle_rj import lorentz_chirp_log_like as like
        s = EnsembleSampler(W, ndims, like, priors, args=[t, y, sigma],
                            tempering_kwargs=dict(ntemps=T), nbranches=len(branch_names), branch_names=branch_names,
                            nleaves_max=nleaves_max, nleaves_min=nleaves_min,
                            moves=GaussianMove(cov) if in_model == "gaussian" else StretchMove(),
                            **({} if rj_moves is None else dict(rj_moves=rj_moves)))
    logp0 = s.compute_log_prior(coords, inds=inds)
    logl0 = s.compute_log_like(coords, inds=inds, logp=logp0)[0]
    state = State(coords, log_like=logl0, log_prior=logp0, inds=inds)

    out = dict(T=T, W=W, ndata=ndata, sigma=float(sigma), nsteps=nsteps, t=t, y=y, nl_max=np.array(nl_max),
               nl_min=np.array(nl_min), cov_factor=float(cov_factor), seed_construct=seed_construct, seed_run=seed_run,
               betas0=np.array(s.temperature_control.betas),
               L0=logl0, P0=logp0, rj_moves="none" if rj_moves is None else rj_moves, in_model=in_model)
    if boxes is None:
        out.update(gauss_box=np.array(GAUSS_BOX), sine_box=np.array(SINE_BOX))
    else:
        out["branch_names"] = np.array(branch_names)
        for k, bx in zip(branch_names, boxes):
            out[f"{k}_box"] = np.array(bx)
    if model != "template":
        out["model"] = model
    for k in branch_names:
        out[f"x0_{k}"], out[f"inds0_{k}"] = coords[k].copy(), inds[k].copy()

    log = []

    def snap(st):
        d = {f"x_{k}": st.branches[k].coords.copy() for k in branch_names}
        d.update({f"inds_{k}": st.branches[k].inds.copy() for k in branch_names})
        d.update(L=st.log_like.copy(), P=st.log_prior.copy())
        return d

    def wrap(move, tag):
        orig = move.propose

        def propose(model, st):
            new, acc = orig(model, st)
            rec = snap(new)
            rec.update(tag=tag, accepted=np.array(acc, copy=True), betas=np.array(s.temperature_control.betas),
                       swaps=np.array(s.temperature_control.swaps_accepted))
            log.append(rec)
            return new, acc
        move.propose = propose

    wrap(s.moves[0], "mh")
    for i, m in enumerate(s.rj_moves if rj_moves is not None else []):
        wrap(m, f"rj{i}")

    np.random.seed(seed_run)                # G for the run
    it = 0
    for st in s.sample(state, iterations=nsteps, store=False):
        assert [r["tag"][:2] for r in log] == (["mh", "rj"] if rj_moves is not None else ["mh"]), [r["tag"] for r in log]
        for r in log:
            pre = f"it{it}_{r['tag'][:2]}_"
            for k, v in r.items():
                if k == "tag":
                    out[pre + "branch"] = int(r["tag"][2:]) if r["tag"].startswith("rj") else -1
                else:
                    out[pre + k] = v
        log.clear()
        it += 1
    out["mh_accepted_total"] = np.array(s.moves[0].accepted)
    if rj_moves is not None:
        out["rj_accepted_total"] = np.stack([np.array(m.accepted) for m in s.rj_moves])
        out["rj_num_proposals"] = np.array([m.num_proposals for m in s.rj_moves])
    path = os.path.join(os.environ.get("GOLDEN_OUT", HERE), name + ".npz")
    np.savez_compressed(path, **out)
    last = "rj" if rj_moves is not None else "mh"
    nl = [int(out[f"it{nsteps - 1}_{last}_inds_{k}"].sum()) for k in branch_names]
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB; in-model accept {out['mh_accepted_total'].mean() / nsteps:.2f}, "
          + (f"rj accept {out['rj_accepted_total'].sum(0).mean() / nsteps:.2f}, " if rj_moves is not None else "") + f"leaves at the end {nl}")


if __name__ == "__main__":
    # the reference test's shape in small: leaf counts free between 0 and the maximum
    capture("rj1_two_branches", T=3, W=8, nl_max=(4, 3), nl_min=(0, 0), nsteps=20)
    # a floor under one branch (edge factors at kmin, rj.py:258-266) and a wider step so that in-model moves are rejected too
    capture("rj2_min_leaves", T=4, W=6, nl_max=(5, 2), nl_min=(1, 0), nsteps=16, cov_factor=4e-3, sigma=1.0, seed_run=99)
    # config-4-like leaf budget (nleaves_max = 10: numpy's 8-way pairwise reduction over the leaf axis of the prior)
    capture("rj3_ten_leaves", T=2, W=6, nl_max=(10, 10), nl_min=(0, 0), nsteps=12, n_init=(4, 2), ndata=60)
    # "iterate_branches" (ensemble.py:434-451; the schedule the reference's own two-branch test runs first, tests/test_eryn.py:
    # 341-507): ONE RJ move per iteration walks through every branch - birth / death, accept, update per branch - then one
    # sweep of swaps without adaptation; its accept mask is the LAST branch's (rj.py:163-388)
    capture("rj4_iterate_branches", T=3, W=8, nl_max=(4, 3), nl_min=(0, 1), nsteps=16, cov_factor=1e-3, seed_run=77,
            rj_moves="iterate_branches")
    # "together" (ensemble.py:414-432): ONE move proposes a birth or death in EVERY branch of a walker at once - all branches'
    # coins and leaf choices first, then the births branch by branch, the factors summed, one accept test (distgenrj.py:150-222)
    capture("rj5_together", T=3, W=8, nl_max=(4, 3), nl_min=(0, 0), nsteps=16, cov_factor=1e-3, seed_run=31,
            rj_moves="together")
    # Round 5: the red / blue StretchMove over a state of several branches and leaves (stretch.py:160-231, red_blue.py:103-330;
    # SURVEY 8 row a4's loop over branches): EVERY leaf slot of EVERY branch moves - one complement walker per branch, one zz per
    # walker, factors (sum of nleaves_max * ndim - 1) log zz - and the leaf masks only decide what the prior and the likelihood see.
    # Fixed masks (no reversible jump; two active leaves per branch: without reversible jump the reference hands a branch's ONLY
    # leaf to the likelihood as a 1-D array, ensemble.py:1438-1441, which its own test model does not take) ...
    capture("rjs1_stretch_fixed_leaves", T=3, W=32, nl_max=(3, 2), nl_min=(0, 0), nsteps=12, rj_moves=None, in_model="stretch",
            seed_run=58, init_spread=4e-4, n_init=(2, 2))
    # ... and as the in-model move beside birth / death (the reference warns, ensemble.py:509-514, and runs it): dead leaves keep
    # moving with the stretch, births land on slots the stretch has been carrying along
    capture("rjs2_stretch_with_rj", T=3, W=32, nl_max=(3, 2), nl_min=(0, 0), nsteps=12, in_model="stretch", seed_run=61,
            init_spread=4e-4)
    # Round 6: reversible jump with a likelihood that is a plain Python function of the packed active leaves (ensemble.py:1306-1334,
    # 1420-1480) - Lorentzian lines + chirps, nothing the device library has a kernel for: the device proposes / tests / updates,
    # the host evaluates (hens_rj_propose / hens_rj_accept)
    capture("rjh1_callable", T=3, W=8, nl_max=(4, 3), nl_min=(0, 0), nsteps=16, model="lorentz_chirp", cov_factor=1e-3, seed_run=412)
    capture("rjh2_callable_together", T=3, W=8, nl_max=(3, 3), nl_min=(1, 0), nsteps=12, model="lorentz_chirp", cov_factor=1e-3,
            seed_run=413, rj_moves="together")
    capture("rjh3_callable_stretch", T=2, W=32, nl_max=(3, 2), nl_min=(0, 0), nsteps=8, model="lorentz_chirp", in_model="stretch",
            seed_run=414, init_spread=4e-4)
    capture("rjh4_callable_iterate", T=3, W=8, nl_max=(4, 3), nl_min=(0, 1), nsteps=12, model="lorentz_chirp", cov_factor=1e-3,
            seed_run=415, rj_moves="iterate_branches")
    # ... and branches of OTHER leaf widths than three (ndims = {ramp: 2, burst: 4}; one branch of one-parameter leaves):
    # separate_branches, "together" with a leaf floor, the stretch move over slots of different widths, "iterate_branches"
    capture("rjn1_widths_2_4", T=3, W=8, nl_max=(3, 3), nl_min=(0, 0), nsteps=16, model="ramp_burst", cov_factor=1e-3, seed_run=511,
            sigma=0.5, n_init=(1, 2))
    capture("rjn2_widths_together", T=3, W=8, nl_max=(2, 4), nl_min=(1, 0), nsteps=16, model="ramp_burst", cov_factor=1e-3, seed_run=512,
            sigma=3.0, n_init=(1, 1), rj_moves="together")
    capture("rjn3_widths_stretch", T=2, W=48, nl_max=(3, 3), nl_min=(0, 0), nsteps=8, model="ramp_burst", in_model="stretch", seed_run=513,
            sigma=0.5, n_init=(1, 2), init_spread=1e-4)
    capture("rjn4_width_1_iterate", T=3, W=8, nl_max=(6,), nl_min=(1,), nsteps=14, model="offset", cov_factor=4e-3, seed_run=514,
            sigma=0.5, n_init=(2,), rj_moves="iterate_branches")
