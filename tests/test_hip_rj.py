"""Reversible-jump leaf packing on the MI355X against the pinned oracle (-m gpu).

Teacher-forced per move (so one knife-edge flip cannot snowball): the oracle (oracle/eryn_oracle_rj.py, pinned bit for
bit to the reference on the three rj* fixtures) records every draw and every intermediate of its iterations; the device
gets the state before a move and the move's draws through the C ABI (hens_rj_mh_step / hens_rj_bd_step / hens_pt_sweep)
and must reproduce: accept masks, leaf masks (``inds``), swap decisions and counts exactly; coordinates and log-prior
exactly; log-likelihood to rtol 1e-12 (device exp / sin / summation order); betas to rtol 1e-13."""
import numpy as np
import pytest

from tests.test_oracle_golden_rj import NAMES, load_rj, make_rj_oracle

pytestmark = pytest.mark.gpu
RTOL_L = 1e-12


def make_engine(fx, o, **kw):
    from eryn_amd.rj import RJEngine, TemplateBranch
    brs = [TemplateBranch(b.name, b.kind, list(zip(b.lo, b.hi)), b.nleaves_max, b.nleaves_min) for b in o.branches]
    return RJEngine(o.T, o.W, brs, fx["t"], fx["y"], float(fx["sigma"]), **kw)


def state_of(rec, prefix, o):
    x = {b.name: rec[f"{prefix}x_{b.name}"] for b in o.branches}
    inds = {b.name: rec[f"{prefix}inds_{b.name}"] for b in o.branches}
    return x, inds, rec[f"{prefix}L"], rec[f"{prefix}P"]


def assert_state(eng, rec, prefix, o, exact_L=False, what=""):
    x, inds, L, P, betas = eng.download()
    ex, einds, eL, eP = state_of(rec, prefix, o)
    for b in o.branches:
        assert np.array_equal(inds[b.name], einds[b.name]), f"{what}: inds of {b.name}"
        assert np.array_equal(x[b.name], ex[b.name]), f"{what}: coordinates of {b.name}"
    assert np.array_equal(P, eP), f"{what}: log-prior"
    if exact_L:
        assert np.array_equal(L, eL), f"{what}: log-like (pure permutation)"
    else:
        np.testing.assert_allclose(L, eL, rtol=RTOL_L, atol=0, err_msg=f"{what}: log-like")
    return betas


def knife(lnpdiff, u):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.abs(lnpdiff - np.log(u)) < 1e-12 * np.maximum(1.0, np.abs(lnpdiff))


@pytest.mark.parametrize("name", NAMES)
def test_rj_moves_match_the_oracle(golden_dir, name):
    fx = load_rj(golden_dir, name)
    o = make_rj_oracle(fx, record=True)
    eng = make_engine(fx, o)
    # initial evaluation (ensemble.py:898-912 with inds)
    eng.upload({b.name: fx[f"x0_{b.name}"] for b in o.branches}, {b.name: fx[f"inds0_{b.name}"] for b in o.branches},
               betas=fx["betas0"])
    eng.eval_state()
    _, _, L, P, _ = eng.download()
    assert np.array_equal(P, fx["P0"])
    np.testing.assert_allclose(L, fx["L0"], rtol=RTOL_L, atol=0)
    n_bd_acc = n_mh_acc = 0
    for it in range(int(fx["nsteps"])):
        o.iteration()
        rec = o.trace[-1]
        what = f"{name} it{it}"
        # ---- in-model move -------------------------------------------------------------------------------------
        x, inds, L, P = state_of(rec, "pre_", o)
        eng.upload(x, inds, L, P, rec["betas_before"])
        eng.set_adapt_time(rec["time_before"])
        steps = {}
        for b in o.branches:                                     # packed draws -> slot layout
            s = np.zeros(x[b.name].shape)
            s[inds[b.name]] = rec["mh_steps"][b.name]
            steps[b.name] = s
        keep = eng.mh_step(steps, rec["mh_u_acc"])
        assert not knife(rec["mh_lnpdiff"], rec["mh_u_acc"]).any()
        assert np.array_equal(keep, rec["mh_accepted"]), f"{what}: in-model accept mask"
        assert_state(eng, rec, "mhupd_", o, what=what + " after the in-model move")
        # swaps + adaptation on the oracle's exact log-likes (decisions depend on them to the last bit)
        eng.upload(*state_of(rec, "mhupd_", o), rec["betas_before"])
        eng.set_adapt_time(rec["time_before"])
        sel, swaps = eng.pt_sweep(rec["iperm"], rec["i1perm"], rec["u_swap"], adapt=True)
        assert np.array_equal(sel, rec["sel"]) and np.array_equal(swaps, rec["swaps"]), f"{what}: swaps"
        betas = assert_state(eng, rec, "mh_", o, exact_L=True, what=what + " after the swaps")
        np.testing.assert_allclose(betas, rec["betas_after"], rtol=1e-13, atol=0)
        # ---- birth / death ---------------------------------------------------------------------------------------
        x, inds, L, P = state_of(rec, "rjpre_", o)
        eng.upload(x, inds, L, P, rec["betas_after"])
        birth = np.zeros((o.T, o.W, 3))
        birth[rec["rj_change"] == +1] = rec["rj_birth"]          # births are listed in (t, w) order (distgenrj.py:85-121)
        keep = eng.bd_step(rec["rj_branch"], rec["rj_change"], rec["rj_leaf"], birth, rec["rj_u_acc"])
        assert not knife(rec["rj_lnpdiff"], rec["rj_u_acc"]).any()
        assert np.array_equal(keep, rec["rj_accepted"]), f"{what}: birth/death accept mask"
        assert_state(eng, rec, "rjupd_", o, what=what + " after birth/death")
        eng.upload(*state_of(rec, "rjupd_", o), rec["betas_after"])
        sel, swaps = eng.pt_sweep(rec["rj_iperm"], rec["rj_i1perm"], rec["rj_u_swap"], adapt=False)   # rj.py:381-382
        assert np.array_equal(sel, rec["rj_sel"]) and np.array_equal(swaps, rec["rj_swaps"])
        betas = assert_state(eng, rec, "rj_", o, exact_L=True, what=what + " after the RJ swaps")
        assert np.array_equal(betas, rec["betas_after"]), "swaps after an RJ move must not adapt the ladder"
        n_bd_acc += int(rec["rj_accepted"].sum())
        n_mh_acc += int(rec["mh_accepted"].sum())
        o.trace.clear()
    assert n_bd_acc > 0 and n_mh_acc > 0
    eng.close()


def test_rj_philox_run_config4_shape():
    """BASELINE config 4 at full size - 2 branches x 10 leaves, ntemps = 8, nwalkers = 2048 - with device-side draws:
    leaf budgets respected, the stored log-like / log-prior are those of the stored leaves (re-evaluation), births and
    deaths both accepted, snapshots NaN-fill unused leaves (backend.py:1049-1059)."""
    from eryn_amd.moves.tempering import make_ladder
    from eryn_amd.rj import RJEngine, TemplateBranch
    T, W, N = 8, 2048, 500
    t = np.linspace(-1, 1, N)
    rs = np.random.RandomState(42)
    gauss_inj = np.array([[3.3, -0.2, 0.1], [2.6, -0.1, 0.1], [3.4, 0.0, 0.1], [2.9, 0.3, 0.1]])
    sine_inj = np.array([[1.3, 10.1, 1.0], [0.8, 4.6, 1.2]])
    y = sum(a * np.exp(-((t - b) ** 2) / (2 * c ** 2)) for a, b, c in gauss_inj) + \
        sum(a * np.sin(2 * np.pi * b * t + c) for a, b, c in sine_inj) + 2.0 * rs.randn(N)
    brs = [TemplateBranch("gauss", "pulse", [(2.5, 3.5), (-1.0, 1.0), (0.01, 0.21)], 10, 0),
           TemplateBranch("sine", "sine", [(0.5, 1.5), (1.0, 20.0), (0.0, 2 * np.pi)], 10, 0)]
    eng = RJEngine(T, W, brs, t, y, 2.0, seed=5)
    x = {"gauss": np.zeros((T, W, 10, 3)), "sine": np.zeros((T, W, 10, 3))}
    inds = {k: np.zeros((T, W, 10), dtype=bool) for k in x}
    for n in range(4):
        x["gauss"][:, :, n] = gauss_inj[n] + 1e-2 * rs.randn(T, W, 3) * [1, 1, 0.1]
        inds["gauss"][:, :, n] = True
    for n in range(2):
        x["sine"][:, :, n] = sine_inj[n] + 1e-2 * rs.randn(T, W, 3)
        inds["sine"][:, :, n] = True
    eng.upload(x, inds, betas=make_ladder(3 * 6, ntemps=T))
    eng.eval_state()
    eng.set_mh_scale(np.full((2, 3), 1e-2) * [[1, 1, 0.1], [1, 1, 1]])
    eng.step(60)
    eng.synchronize()
    x1, inds1, L1, P1, betas = eng.download(nan_fill=True)
    c = eng.counters()
    assert c["num_mh"] == 60 and c["num_bd"] == 60
    assert c["accepted_mh"].sum() > 0 and c["accepted_bd"].sum() > 0 and c["swaps_total"].sum() > 0
    for k in x1:
        nl = inds1[k].sum(axis=-1)
        assert nl.min() >= 0 and nl.max() <= 10
        assert np.isnan(x1[k][~inds1[k]]).all() and np.isfinite(x1[k][inds1[k]]).all()
    assert np.any(inds1["gauss"].sum(-1) != 4) or np.any(inds1["sine"].sum(-1) != 2)      # the model dimension moved
    assert np.isfinite(P1).all() and betas[0] == 1.0 and np.all(np.diff(betas) < 0)
    eng.upload(x1, inds1, betas=betas)                          # NaN-filled snapshot -> records again
    eng.eval_state()
    _, _, L2, P2, _ = eng.download()
    np.testing.assert_allclose(L2, L1, rtol=1e-12, atol=0)
    assert np.array_equal(P2, P1)
    eng.close()


@pytest.mark.parametrize("name", NAMES)
def test_rj_sampler_reproduces_the_reference_chain(golden_dir, name):
    """The sampler-level mirror (eryn_amd.rj.RJEnsembleSampler, the reference's constructor contract for this path)
    free-running with the reference's seeds lands on the reference's chain: every leaf slot (dead ones included) and
    log-prior exact, leaf masks exact, log-like rtol 1e-12, betas rtol 1e-13, accept counters exact."""
    from eryn_amd.prior import uniform_dist
    from eryn_amd.rj import GaussianLeafMove, RJEnsembleSampler, TemplateLikelihood
    from eryn_amd.state import State
    fx = load_rj(golden_dir, name)
    names = ["gauss", "sine"]
    n = int(fx["nsteps"])
    priors = {"gauss": {i: uniform_dist(*fx["gauss_box"][i]) for i in range(3)},
              "sine": {i: uniform_dist(*fx["sine_box"][i]) for i in range(3)}}
    cov = {k: np.diag(np.ones(3)) * float(fx["cov_factor"]) for k in names}
    np.random.seed(int(fx["seed_construct"]))          # R := snapshot of the global stream at construction
    s = RJEnsembleSampler(int(fx["W"]), {k: 3 for k in names}, TemplateLikelihood({"gauss": "pulse", "sine": "sine"},
                          fx["t"], fx["y"], float(fx["sigma"])), priors, tempering_kwargs=dict(ntemps=int(fx["T"])),
                          nbranches=2, branch_names=names, nleaves_max=dict(zip(names, map(int, fx["nl_max"]))),
                          nleaves_min=dict(zip(names, map(int, fx["nl_min"]))), moves=GaussianLeafMove(cov),
                          rj_moves="separate_branches")
    assert np.array_equal(s.temperature_control.betas, fx["betas0"])
    coords = {k: fx[f"x0_{k}"] for k in names}
    inds = {k: fx[f"inds0_{k}"] for k in names}
    logp = s.compute_log_prior(coords, inds=inds)
    logl = s.compute_log_like(coords, inds=inds, logp=logp)[0]
    assert np.array_equal(logp, fx["P0"])
    np.testing.assert_allclose(logl, fx["L0"], rtol=RTOL_L, atol=0)
    np.random.seed(int(fx["seed_run"]))
    last = s.run_mcmc(State(coords, log_like=fx["L0"], log_prior=logp, inds=inds), n, store=True)
    pre = f"it{n - 1}_rj_"
    for k in names:
        assert np.array_equal(last.branches[k].inds, fx[pre + f"inds_{k}"]), f"inds of {k}"
        assert np.array_equal(last.branches[k].coords, fx[pre + f"x_{k}"]), f"coordinates of {k}"
    assert np.array_equal(last.log_prior, fx[pre + "P"])
    np.testing.assert_allclose(last.log_like, fx[pre + "L"], rtol=RTOL_L, atol=0)
    np.testing.assert_allclose(last.betas, fx[pre + "betas"], rtol=1e-13, atol=0)
    assert np.array_equal(s.moves[0].accepted, fx["mh_accepted_total"])
    assert np.array_equal(np.stack(s.rj_accepted), fx["rj_accepted_total"])
    assert np.array_equal(np.array(s.rj_num_proposals), fx["rj_num_proposals"])
    # stored steps carry the reference's NaN fill of unused leaves (backend.py:1049-1059)
    assert len(s.chain) == n
    mid = s.chain[n // 2]
    for k in names:
        assert np.array_equal(mid.branches[k].inds, fx[f"it{n // 2}_rj_inds_{k}"])
        assert np.isnan(mid.branches[k].coords[~mid.branches[k].inds]).all()
    s.engine.close()


def test_rj_sampler_philox_mode():
    from eryn_amd.prior import uniform_dist
    from eryn_amd.rj import GaussianLeafMove, RJEnsembleSampler, TemplateLikelihood
    from eryn_amd.state import State
    T, W, N = 4, 64, 100
    t = np.linspace(-1, 1, N)
    rs = np.random.RandomState(3)
    y = 3.0 * np.exp(-((t - 0.1) ** 2) / (2 * 0.1 ** 2)) + 1.0 * np.sin(2 * np.pi * 5.0 * t + 1.0) + 1.5 * rs.randn(N)
    names = ["gauss", "sine"]
    priors = {"gauss": {0: uniform_dist(2.5, 3.5), 1: uniform_dist(-1, 1), 2: uniform_dist(0.01, 0.21)},
              "sine": {0: uniform_dist(0.5, 1.5), 1: uniform_dist(1.0, 20.0), 2: uniform_dist(0.0, 2 * np.pi)}}
    s = RJEnsembleSampler(W, {k: 3 for k in names}, TemplateLikelihood({"gauss": "pulse", "sine": "sine"}, t, y, 1.5), priors,
                          tempering_kwargs=dict(ntemps=T), branch_names=names, nleaves_max={"gauss": 4, "sine": 3},
                          moves=GaussianLeafMove({k: np.eye(3) * 1e-4 for k in names}), rng="philox", seed=8)
    coords = {"gauss": np.zeros((T, W, 4, 3)), "sine": np.zeros((T, W, 3, 3))}
    inds = {"gauss": np.zeros((T, W, 4), dtype=bool), "sine": np.zeros((T, W, 3), dtype=bool)}
    coords["gauss"][:, :, 0] = [3.0, 0.1, 0.1]
    coords["sine"][:, :, 0] = [1.0, 5.0, 1.0]
    inds["gauss"][:, :, 0] = inds["sine"][:, :, 0] = True
    last = s.run_mcmc(State(coords, inds=inds), 30, burn=5, thin_by=2, store=True)
    assert len(s.chain) == 30 and s.iteration == 35
    nl = s.get_nleaves()
    assert nl["gauss"].shape == (30, T, W) and nl["gauss"].max() <= 4 and nl["sine"].max() <= 3
    assert np.isfinite(last.log_like).all() and (nl["gauss"][-1] != 1).any()
    s.engine.close()
