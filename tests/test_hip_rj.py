"""Reversible-jump leaf packing on the MI355X against the pinned oracle (-m gpu).

Teacher-forced per move (so one knife-edge flip cannot snowball): the oracle (oracle/eryn_oracle_rj.py, pinned bit for
bit to the reference on the three rj* fixtures) records every draw and every intermediate of its iterations; the device
gets the state before a move and the move's draws through the C ABI (hens_rj_mh_step / hens_rj_bd_step / hens_pt_sweep)
and must reproduce: accept masks, leaf masks (``inds``), swap decisions and counts exactly; coordinates and log-prior
exactly; log-likelihood to rtol 1e-12 (device exp / sin / summation order); betas to rtol 1e-13."""
import numpy as np
import pytest

from tests import tolerance_log as tol
from tests.test_oracle_golden_rj import NAMES, NAMES_ALL, NAMES_STRETCH, load_rj, make_rj_oracle

pytestmark = pytest.mark.gpu
# the template likelihood sums 500 data points of FP64 exp / sin evaluated by the device's math library (each within an ulp or two
# of libm's, not bit-equal) and squares a residual that nearly cancels: 1e-12 here, against 1e-13 for the Gaussian quadratic form
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
        tol.check_logl(L, eL, RTOL_L, f"{what}: log-like")
    return betas


def knife(lnpdiff, u):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.abs(lnpdiff - np.log(u)) < 1e-12 * np.maximum(1.0, np.abs(lnpdiff))


@pytest.mark.parametrize("name", NAMES_ALL)
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
    tol.check_logl(L, fx["L0"], RTOL_L, 'template log-like')
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
        # ---- birth / death: on one branch, or ("iterate_branches") on every branch in turn ------------------------------
        for sub in rec.get("rj_sub", [rec]):
            x, inds, L, P = state_of(sub, "rjpre_", o)
            eng.upload(x, inds, L, P, rec["betas_after"])
            if "rj_branches" in sub:                             # "together": every branch in one proposal
                nb = len(o.branches)
                birth = np.zeros((nb, o.T, o.W, 3))
                for bi in range(nb):
                    birth[bi][sub["rj_change_all"][bi] == +1] = sub["rj_birth_all"][bi]
                keep = eng.bd_all_step(np.stack(sub["rj_change_all"]), np.stack(sub["rj_leaf_all"]), birth, sub["rj_u_acc"])
                assert not knife(sub["rj_lnpdiff"], sub["rj_u_acc"]).any()
                assert np.array_equal(keep, sub["rj_accepted"]), f"{what}: birth/death accept mask (all branches)"
                assert_state(eng, sub, "rjupd_", o, what=what + " after birth/death on all branches")
                continue
            birth = np.zeros((o.T, o.W, 3))
            birth[sub["rj_change"] == +1] = sub["rj_birth"]      # births are listed in (t, w) order (distgenrj.py:85-121)
            keep = eng.bd_step(sub["rj_branch"], sub["rj_change"], sub["rj_leaf"], birth, sub["rj_u_acc"])
            assert not knife(sub["rj_lnpdiff"], sub["rj_u_acc"]).any()
            assert np.array_equal(keep, sub["rj_accepted"]), f"{what}: birth/death accept mask (branch {sub['rj_branch']})"
            assert_state(eng, sub, "rjupd_", o, what=what + f" after birth/death on branch {sub['rj_branch']}")
        eng.upload(*state_of(sub, "rjupd_", o), rec["betas_after"])
        sel, swaps = eng.pt_sweep(rec["rj_iperm"], rec["rj_i1perm"], rec["rj_u_swap"], adapt=False)   # rj.py:381-382
        assert np.array_equal(sel, rec["rj_sel"]) and np.array_equal(swaps, rec["rj_swaps"])
        betas = assert_state(eng, rec, "rj_", o, exact_L=True, what=what + " after the RJ swaps")
        assert np.array_equal(betas, rec["betas_after"]), "swaps after an RJ move must not adapt the ladder"
        n_bd_acc += int(rec["rj_accepted"].sum())
        n_mh_acc += int(rec["mh_accepted"].sum())
        o.trace.clear()
    assert n_bd_acc > 0 and n_mh_acc > 0
    eng.close()


def test_rj_chain_resumed_from_a_downloaded_state_is_the_uninterrupted_chain():
    """The production path keeps every walker's model resident and evaluates birth / death as `model +- one leaf`; templates and
    log-likelihoods are re-evaluated from the coordinates whenever the state has crossed the C ABI since the last call - upload,
    parity move, DOWNLOAD - (and when iteration % 64 == 63), so the state behind a download is a function of what the download
    returned: the chain continues the same way in the same context or in a NEW one after an upload (round 4 rebuilt the
    templates only, and only after an upload: the resumed chain's last bits then differed)."""
    from eryn_amd.moves.tempering import make_ladder
    from eryn_amd.rj import RJEngine, TemplateBranch
    T, W, N = 4, 256, 500
    t = np.linspace(-1, 1, N)
    rs = np.random.RandomState(42)
    gauss_inj = np.array([[3.3, -0.2, 0.1], [2.6, -0.1, 0.1], [3.4, 0.0, 0.1]])
    sine_inj = np.array([[1.3, 10.1, 1.0], [0.8, 4.6, 1.2]])
    y = sum(a * np.exp(-((t - b) ** 2) / (2 * c ** 2)) for a, b, c in gauss_inj) + \
        sum(a * np.sin(2 * np.pi * b * t + c) for a, b, c in sine_inj) + 2.0 * rs.randn(N)
    brs = [TemplateBranch("gauss", "pulse", [(2.5, 3.5), (-1.0, 1.0), (0.01, 0.21)], 6, 0),
           TemplateBranch("sine", "sine", [(0.5, 1.5), (1.0, 20.0), (0.0, 2 * np.pi)], 6, 0)]
    x = {"gauss": np.zeros((T, W, 6, 3)), "sine": np.zeros((T, W, 6, 3))}
    inds = {k: np.zeros((T, W, 6), dtype=bool) for k in x}
    for n in range(3):
        x["gauss"][:, :, n] = gauss_inj[n] + 1e-2 * rs.randn(T, W, 3) * [1, 1, 0.1]
        inds["gauss"][:, :, n] = True
    for n in range(2):
        x["sine"][:, :, n] = sine_inj[n] + 1e-2 * rs.randn(T, W, 3)
        inds["sine"][:, :, n] = True
    scale = np.full((2, 3), 1e-2) * [[1, 1, 0.1], [1, 1, 1]]
    calls = (70, 5, 90)                                # (the calls cross iteration % 64 == 63 at different offsets)

    def fresh():
        e = RJEngine(T, W, brs, t, y, 2.0, seed=5)
        e.set_mh_scale(scale)
        return e

    a = fresh()
    a.upload(x, inds, betas=make_ladder(3 * 5, ntemps=T))
    a.eval_state()
    snaps = []
    for n in calls:
        a.step(n)
        snaps.append((a.download(), a.iteration(), a.counters()["adapt_time"]))
    a.close()
    (x1, i1, L1, P1, b1), it1, at1 = snaps[0]
    b = fresh()                                        # a NEW context continues from the first download
    b.upload(x1, i1, L1, P1, b1)
    b.eng.set_iteration(it1)
    b.set_adapt_time(at1)
    for k, n in enumerate(calls[1:], start=1):
        b.step(n)
        xb, ib, Lb, Pb, bb = b.download()
        (xa, ia, La, Pa, ba), ita, _ = snaps[k]
        assert b.iteration() == ita
        for name in xa:
            assert np.array_equal(ia[name], ib[name]), f"leaf masks differ after call {k}"
            assert np.array_equal(xa[name], xb[name]), f"coordinates differ after call {k}"
        assert np.array_equal(La, Lb) and np.array_equal(Pa, Pb) and np.array_equal(ba, bb), f"log-probabilities / ladder differ after call {k}"
    b.close()


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


@pytest.mark.parametrize("name", NAMES_ALL)
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
                          rj_moves=str(fx["rj_moves"]))
    assert np.array_equal(s.temperature_control.betas, fx["betas0"])
    coords = {k: fx[f"x0_{k}"] for k in names}
    inds = {k: fx[f"inds0_{k}"] for k in names}
    logp = s.compute_log_prior(coords, inds=inds)
    logl = s.compute_log_like(coords, inds=inds, logp=logp)[0]
    assert np.array_equal(logp, fx["P0"])
    tol.check_logl(logl, fx["L0"], RTOL_L, 'template log-like')
    np.random.seed(int(fx["seed_run"]))
    last = s.run_mcmc(State(coords, log_like=fx["L0"], log_prior=logp, inds=inds), n, store=True)
    pre = f"it{n - 1}_rj_"
    for k in names:
        assert np.array_equal(last.branches[k].inds, fx[pre + f"inds_{k}"]), f"inds of {k}"
        assert np.array_equal(last.branches[k].coords, fx[pre + f"x_{k}"]), f"coordinates of {k}"
    assert np.array_equal(last.log_prior, fx[pre + "P"])
    tol.check_logl(last.log_like, fx[pre + "L"], RTOL_L, 'template log-like')
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


@pytest.mark.parametrize("name", NAMES_STRETCH)
def test_stretch_move_over_branches_and_leaves_matches_the_oracle(golden_dir, name):
    """Round 5 (SURVEY 8 row a4 over several branches and leaves): the red / blue StretchMove on leaf-packing records,
    hens_rj_stretch_split, teacher-forced per half against the oracle (pinned bit for bit on the rjs* fixtures captured from the
    reference): accept masks exact and no knife edge, every leaf slot - dead ones move too - and the log-prior exact, leaf masks
    untouched, log-like 1e-12; then swaps + adaptation, and (rjs2) the birth / death move on the state the stretch left."""
    fx = load_rj(golden_dir, name)
    o = make_rj_oracle(fx, record=True)
    eng = make_engine(fx, o)
    n_acc = 0
    for it in range(int(fx["nsteps"])):
        o.iteration()
        rec = o.trace[-1]
        what = f"{name} it{it}"
        x, inds, L, P = state_of(rec, "pre_", o)
        eng.upload(x, inds, L, P, rec["betas_before"])
        eng.set_adapt_time(rec["time_before"])
        for split in range(2):
            keep = eng.stretch_split(split, rec["st_labels"], rec[f"st_rint{split}"], rec[f"st_u_zz{split}"], rec[f"st_u_acc{split}"])
            assert not knife(rec[f"st_lnpdiff{split}"], rec[f"st_u_acc{split}"]).any()
            assert np.array_equal(keep, rec[f"st_keep{split}"]), f"{what}: accept mask of half {split}"
            assert_state(eng, rec, f"stupd{split}_", o, what=what + f" after half {split}")
            n_acc += int(keep.sum())
        # swaps + adaptation on the oracle's exact log-likes (decisions depend on them to the last bit)
        eng.upload(*state_of(rec, "mhupd_", o), rec["betas_before"])
        eng.set_adapt_time(rec["time_before"])
        sel, swaps = eng.pt_sweep(rec["iperm"], rec["i1perm"], rec["u_swap"], adapt=True)
        assert np.array_equal(sel, rec["sel"]) and np.array_equal(swaps, rec["swaps"]), f"{what}: swaps"
        betas = assert_state(eng, rec, "mh_", o, exact_L=True, what=what + " after the swaps")
        np.testing.assert_allclose(betas, rec["betas_after"], rtol=1e-13, atol=0)
        if o.schedule != "none":                                 # the birth / death move on what the stretch left behind
            x, inds, L, P = state_of(rec, "rjpre_", o)
            eng.upload(x, inds, L, P, rec["betas_after"])
            birth = np.zeros((o.T, o.W, 3))
            birth[rec["rj_change"] == +1] = rec["rj_birth"]
            keep = eng.bd_step(rec["rj_branch"], rec["rj_change"], rec["rj_leaf"], birth, rec["rj_u_acc"])
            assert np.array_equal(keep, rec["rj_accepted"]), f"{what}: birth/death accept mask"
            assert_state(eng, rec, "rjupd_", o, what=what + " after birth/death")
    assert n_acc > 0
    eng.close()


@pytest.mark.parametrize("name", NAMES_STRETCH)
def test_rj_sampler_with_the_stretch_move_reproduces_the_reference_chain(golden_dir, name):
    """RJEnsembleSampler(moves=StretchLeafMove(), rj_moves=None | "separate_branches") free-running from the reference's two seeds
    lands on the reference's chain: every leaf slot and the log-prior exact, leaf masks exact, log-like 1e-12, ladder 1e-13."""
    from eryn_amd.prior import uniform_dist
    from eryn_amd.rj import RJEnsembleSampler, StretchLeafMove, TemplateLikelihood
    from eryn_amd.state import State
    fx = load_rj(golden_dir, name)
    names = ["gauss", "sine"]
    n = int(fx["nsteps"])
    rj = None if str(fx["rj_moves"]) == "none" else str(fx["rj_moves"])
    priors = {"gauss": {i: uniform_dist(*fx["gauss_box"][i]) for i in range(3)},
              "sine": {i: uniform_dist(*fx["sine_box"][i]) for i in range(3)}}
    np.random.seed(int(fx["seed_construct"]))          # R := snapshot of the global stream at construction
    s = RJEnsembleSampler(int(fx["W"]), {k: 3 for k in names}, TemplateLikelihood({"gauss": "pulse", "sine": "sine"},
                          fx["t"], fx["y"], float(fx["sigma"])), priors, tempering_kwargs=dict(ntemps=int(fx["T"])),
                          nbranches=2, branch_names=names, nleaves_max=dict(zip(names, map(int, fx["nl_max"]))),
                          nleaves_min=dict(zip(names, map(int, fx["nl_min"]))), moves=StretchLeafMove(), rj_moves=rj)
    coords = {k: fx[f"x0_{k}"] for k in names}
    inds = {k: fx[f"inds0_{k}"] for k in names}
    np.random.seed(int(fx["seed_run"]))
    last = s.run_mcmc(State(coords, log_like=fx["L0"], log_prior=fx["P0"], inds=inds), n, store=False)
    pre = f"it{n - 1}_{'mh' if rj is None else 'rj'}_"
    for k in names:
        assert np.array_equal(last.branches[k].inds, fx[pre + f"inds_{k}"]), f"inds of {k}"
        assert np.array_equal(last.branches[k].coords, fx[pre + f"x_{k}"]), f"coordinates of {k}"
    assert np.array_equal(last.log_prior, fx[pre + "P"])
    tol.check_logl(last.log_like, fx[pre + "L"], RTOL_L, 'template log-like')
    np.testing.assert_allclose(last.betas, fx[pre + "betas"], rtol=1e-13, atol=0)
    assert np.array_equal(s.moves[0].accepted, fx["mh_accepted_total"])
    if rj is not None:
        assert np.array_equal(np.stack(s.rj_accepted), fx["rj_accepted_total"])
    s.engine.close()


def test_stretch_move_on_records_refuses_too_few_walkers():
    """red_blue.py:103-114 counts every leaf slot of every branch: nwalkers < 2 * sum(nleaves_max * ndim) raises RuntimeError."""
    from eryn_amd.rj import RJEngine, TemplateBranch
    t = np.linspace(-1, 1, 20)
    brs = [TemplateBranch("gauss", "pulse", [(2.5, 3.5), (-1.0, 1.0), (0.01, 0.21)], 3, 0),
           TemplateBranch("sine", "sine", [(0.5, 1.5), (1.0, 20.0), (0.0, 2 * np.pi)], 2, 0)]
    T, W = 2, 16                                                   # 15 leaf coordinates: 30 walkers needed
    eng = RJEngine(T, W, brs, t, np.zeros(20), 1.0)
    x = {"gauss": np.zeros((T, W, 3, 3)), "sine": np.zeros((T, W, 2, 3))}
    inds = {"gauss": np.zeros((T, W, 3), dtype=bool), "sine": np.zeros((T, W, 2), dtype=bool)}
    eng.upload(x, inds, betas=np.array([1.0, 0.5]))
    eng.eval_state()
    labels = np.tile(np.arange(W) % 2, (T, 1)).astype(np.uint8)
    with pytest.raises(RuntimeError):
        eng.stretch_split(0, labels, np.zeros((2, T, W // 2), dtype=np.int64), np.full((T, W // 2), 0.5), np.full((T, W // 2), 0.5))
    eng.close()
    # ... unless the move lives dangerously (red_blue.py:108: the guard is skipped), as StretchMove(live_dangerously=True) does upstream
    eng = RJEngine(T, W, brs, t, np.zeros(20), 1.0, live_dangerously=True)
    eng.upload(x, inds, betas=np.array([1.0, 0.5]))
    eng.eval_state()
    keep = eng.stretch_split(0, labels, np.zeros((2, T, W // 2), dtype=np.int64), np.full((T, W // 2), 0.5), np.full((T, W // 2), 0.5))
    assert keep.shape == (T, W // 2)
    eng.close()


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


# ---- the PRODUCTION path (hens_rj_step: what bench.py --workload cfg4 times) under the oracle -------------------------------
def _replay_oracle_class():
    from oracle import eryn_oracle_rj as orj

    class ReplayRJ(orj.OracleRJSampler):
        """The pinned RJ oracle with its draw SOURCES replaced by what hens_rj_debug_draws exports for an iteration - the
        arithmetic, the order of operations and every decision stay the oracle's."""

        def load(self, d, offsets):
            self.d, self.offsets = d, offsets

        def _draw_move_choice(self):
            pass                                                     # (one in-model move: the device draws nothing for it)

        def _draw_steps(self, b, n):
            tt, ww, ll = np.where(self.st.inds[b.name])              # packed active leaves, (t, w, leaf) order
            assert len(tt) == n
            idx = self.offsets[b.name] + ll[:, None] * 3 + np.arange(3)
            return self.d["step"][tt[:, None], ww[:, None], idx]

        def _k(self):                                                # entry of the birth / death arrays' branch axis
            if self.schedule == "separate_branches":
                return 0
            return min(self._bi, len(self.branches) - 1)             # ("together": _bi = nbranches while the ONE accept uniform is drawn)

        def _rj_branch(self, bi, rec=None):
            self._bi = bi
            return super()._rj_branch(bi, rec)

        def _draw_accept(self, which):
            return self.d["u_mh"] if which == "mh" else self.d["u_bd"][self._k()]

        def _draw_branch(self, nb):
            if self.schedule != "separate_branches":                 # (the choice among ONE move: nothing to draw on the device)
                assert nb == 1 and self.d["branch"] == -1 and self.d["coin"].shape[0] == len(self.branches)
                return 0
            assert 0 <= self.d["branch"] < nb and self.d["coin"].shape[0] == 1
            return self.d["branch"]

        def _draw_coin(self, shape):
            c = self.d["coin"][self._k()].astype(np.int64)
            assert c.shape == shape and set(np.unique(c)) <= {-1, 1}
            return c

        def _draw_leaf(self, tt, w, candidates):                     # uniform over the candidates, ascending slot order
            return candidates[(int(self.d["sel"][self._k()][tt, w]) * len(candidates)) >> 32]

        def _draw_birth(self, b, bt, bw):
            return self.d["birth"][self._k()][bt, bw]

        def _pt(self, adapt, rec):
            self._casc = "mh" if adapt else "bd"                     # the cascade after the in-model move adapts (rj.py:381-382)
            return super()._pt(adapt, rec)

        def _draw_pair(self, j, W):
            slot, u = self.d["slot_" + self._casc].astype(np.int64), self.d["uswap_" + self._casc]
            i = self.T - 1 - j                                       # pair (i, i-1): column c meets slot[i][c] and slot[i-1][c]
            for row in (slot[i], slot[i - 1]):
                assert np.array_equal(np.sort(row), np.arange(W)), "a rung's column map must be a permutation"
            return slot[i], slot[i - 1], u[j]

    return ReplayRJ


def _replay_rj(T, W, nl_max, nl_min, ndata, iters, seed, start_leaves=(2, 1), calls=None, schedule="separate_branches"):
    from oracle import eryn_oracle_rj as orj
    from eryn_amd.moves.tempering import make_ladder
    from eryn_amd.rj import RJEngine, TemplateBranch
    rs = np.random.RandomState(seed)
    t = np.linspace(-1, 1, ndata)
    gauss_inj = np.array([[3.3, -0.2, 0.1], [2.6, -0.1, 0.1], [3.4, 0.0, 0.1], [2.9, 0.3, 0.1]])
    sine_inj = np.array([[1.3, 10.1, 1.0], [0.8, 4.6, 1.2]])
    sigma = 2.0
    y = sum(a * np.exp(-((t - b) ** 2) / (2 * c ** 2)) for a, b, c in gauss_inj) + \
        sum(a * np.sin(2 * np.pi * b * t + c) for a, b, c in sine_inj) + sigma * rs.randn(ndata)
    boxes = {"gauss": [(2.5, 3.5), (-1.0, 1.0), (0.01, 0.21)], "sine": [(0.5, 1.5), (1.0, 20.0), (0.0, 2 * np.pi)]}
    kinds = {"gauss": "pulse", "sine": "sine"}
    scale = np.array([[1e-2, 1e-2, 1e-3], [1e-2, 1e-2, 1e-2]])
    names = ["gauss", "sine"]
    brs = [TemplateBranch(k, kinds[k], boxes[k], nl_max[i], nl_min[i]) for i, k in enumerate(names)]
    eng = RJEngine(T, W, brs, t, y, sigma, seed=seed)
    x = {k: np.zeros((T, W, nl_max[i], 3)) for i, k in enumerate(names)}
    inds = {k: np.zeros((T, W, nl_max[i]), dtype=bool) for i, k in enumerate(names)}
    inj = {"gauss": gauss_inj, "sine": sine_inj}
    for i, k in enumerate(names):
        for n in range(min(start_leaves[i], nl_max[i])):
            x[k][:, :, n] = inj[k][n % len(inj[k])] + 1e-2 * rs.randn(T, W, 3) * [1, 1, 0.1 if k == "gauss" else 1]
            inds[k][:, :, n] = True
    betas0 = make_ladder(3 * sum(start_leaves), ntemps=T)
    eng.upload(x, inds, betas=betas0)
    eng.eval_state()
    eng.set_mh_scale(scale)
    eng.set_schedule(schedule)
    x0, inds0, L0, P0, _ = eng.download()
    okind = {"pulse": orj.KIND_PULSE, "sine": orj.KIND_SINE}
    obr = [orj.Branch(k, okind[kinds[k]], boxes[k], nl_max[i], nl_min[i], cov=np.diag(scale[i] ** 2)) for i, k in enumerate(names)]
    o = _replay_oracle_class()(obr, x0, inds0, t, y, sigma, None, None, betas0, schedule=schedule)
    assert np.array_equal(o.st.P, P0)
    tol.check_logl(L0, o.st.L, RTOL_L, 'template log-like')
    offsets = {b.name: eng.off[i] for i, b in enumerate(brs)}
    mh_acc, bd_acc, swaps_total, nbd, done = np.zeros((T, W)), np.zeros((T, W)), np.zeros(T - 1), [0, 0], 0
    for n in (calls or (iters,)):
        it0 = eng.iteration()
        eng.step(n)
        eng.synchronize()
        for it in range(it0, it0 + n):
            o.load(eng.debug_draws(it), offsets)
            acc, bi, racc = o.iteration()
            mh_acc += acc
            bd_acc += racc                                           # ("iterate_branches": the last branch's mask, like the device)
            if schedule != "separate_branches":
                nbd = [n_ + 1 for n_ in nbd]
            else:
                nbd[bi] += 1
        done += n
        x1, inds1, L1, P1, betas1 = eng.download()
        what = f"hens_rj_step vs oracle after {done} iterations"
        for k in names:
            assert np.array_equal(inds1[k], o.st.inds[k]), f"{what}: leaf masks of {k}"
            assert np.array_equal(x1[k], o.st.x[k]), f"{what}: coordinates of {k} (dead slots included)"
        assert np.array_equal(P1, o.st.P), f"{what}: log-prior"
        tol.check_logl(L1, o.st.L, RTOL_L, what)
        np.testing.assert_allclose(betas1, o.st.betas, rtol=1e-13, atol=0, err_msg=what)
        c = eng.counters()
        assert np.array_equal(c["accepted_mh"], mh_acc) and np.array_equal(c["accepted_bd"], bd_acc), f"{what}: accept counters"
        assert c["num_mh"] == done and c["num_bd"] == done
        assert np.array_equal(c["swaps_last"], o.swaps_accepted), f"{what}: swap counts of the last cascade"
    assert mh_acc.sum() > 0 and bd_acc.sum() > 0 and min(nbd) > 0, "both moves, both branches and both outcomes must occur"
    eng.close()
    return o


@pytest.mark.parametrize("T,W,nl_max,nl_min,iters", [(4, 10, (3, 4), (0, 0), 8), (4, 10, (3, 4), (1, 0), 8), (3, 12, (10, 10), (0, 0), 6),
                                                      (2, 64, (2, 2), (0, 1), 6)])
def test_rj_production_step_replayed_through_the_oracle_small(T, W, nl_max, nl_min, iters):
    """hens_rj_step - branch choice, in-model steps, birth / death coins, leaf choice, birth draws, accept uniforms, both
    cascades and the adaptation - on the three rj fixtures' shapes (free leaf counts, a floor under one branch, a 10-leaf
    budget) and a tight budget whose edge rule fires all the time.  A wrong branch, a mis-keyed birth draw or a leaf picked
    from the wrong candidate list fails here."""
    _replay_rj(T, W, nl_max, nl_min, ndata=60, iters=iters, seed=11, start_leaves=(2, 1), calls=(3, iters - 3))


@pytest.mark.parametrize("T,W,nl_max,nl_min,iters", [(3, 8, (4, 3), (0, 1), 8), (2, 64, (2, 2), (0, 0), 6)])
def test_rj_production_step_iterate_branches_replayed_through_the_oracle(T, W, nl_max, nl_min, iters):
    """rj_moves="iterate_branches" (hens_rj_set_schedule 1): every branch's birth / death move in turn within one iteration, each
    with its own draws (the branch is part of the Philox key), one sweep of swaps after the last, accept counts the last
    branch's - on the rj4 fixture's shape and on a tight budget."""
    _replay_rj(T, W, nl_max, nl_min, ndata=60, iters=iters, seed=13, start_leaves=(2, 1), calls=(3, iters - 3),
               schedule="iterate_branches")


@pytest.mark.parametrize("T,W,nl_max,nl_min,iters", [(3, 8, (4, 3), (0, 0), 8), (2, 64, (2, 3), (0, 1), 14)])
def test_rj_production_step_together_replayed_through_the_oracle(T, W, nl_max, nl_min, iters):
    """rj_moves="together" (hens_rj_set_schedule 2): one proposal changes a leaf in every branch of a walker - per-branch coins,
    leaf choices and births, the factors summed, one accept uniform."""
    _replay_rj(T, W, nl_max, nl_min, ndata=60, iters=iters, seed=17, start_leaves=(2, 2), calls=(3, iters - 3), schedule="together")


@pytest.mark.parametrize("T,W,nl_max,nl_min,iters,ndata,schedule", [
    (3, 12, (4, 3), (0, 0), 8, 130, "separate_branches"), (3, 8, (4, 3), (0, 1), 8, 130, "iterate_branches"),
    (3, 8, (4, 3), (0, 0), 8, 130, "together"), (2, 16, (12, 12), (0, 0), 6, 60, "separate_branches"),
    (2, 16, (12, 12), (0, 0), 6, 130, "separate_branches"), (2, 16, (12, 11), (0, 0), 6, 200, "together")])
def test_rj_production_step_uniform_grid_and_wide_records(T, W, nl_max, nl_min, iters, ndata, schedule):
    """Round 5: the production kernels' uniform-grid likelihood (more than 64 data points on a linspace: a lane owns 8 consecutive
    points, pulses by recurrence, sines by rotation; pulses narrower than the grid step - c < 2 / (ndata - 1) is inside the prior -
    take the exp per point) under every schedule, birth / death by difference with several branches changing at once
    ("together"), and records of more than 64 coordinates (2 x 12 leaves x 3 = 72: the second pass of the per-coordinate
    phases) - replayed through the oracle like the small shapes above."""
    _replay_rj(T, W, nl_max, nl_min, ndata=ndata, iters=iters, seed=19, start_leaves=(2, 2), calls=(3, iters - 3), schedule=schedule)


def test_rj_production_step_replayed_through_the_oracle_config4():
    """BASELINE config 4 at full size (8 x 2048 walkers, 2 branches x 10 leaves, 500 data points): three iterations of
    hens_rj_step replayed."""
    _replay_rj(8, 2048, (10, 10), (0, 0), ndata=500, iters=3, seed=5, start_leaves=(4, 2))


_FOLD_WORKER = r"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from eryn_amd.moves.tempering import make_ladder
from eryn_amd.rj import RJEngine, TemplateBranch
T, W, N, NL = 6, 256, 200, 5
t = np.linspace(-1, 1, N); rs = np.random.RandomState(3)
y = 3.0 * np.exp(-((t + 0.2) ** 2) / 0.02) + 1.2 * np.sin(2 * np.pi * 7.3 * t + 1.0) + 2.0 * rs.randn(N)
brs = [TemplateBranch("gauss", "pulse", [(2.5, 3.5), (-1.0, 1.0), (0.01, 0.21)], NL, 0),
       TemplateBranch("sine", "sine", [(0.5, 1.5), (1.0, 20.0), (0.0, 2 * np.pi)], NL, 0)]
eng = RJEngine(T, W, brs, t, y, 2.0, seed=9)
x = {"gauss": np.zeros((T, W, NL, 3)), "sine": np.zeros((T, W, NL, 3))}
inds = {k: np.zeros((T, W, NL), dtype=bool) for k in x}
x["gauss"][:, :, 0] = [3.0, -0.2, 0.1] + 1e-2 * rs.randn(T, W, 3) * [1, 1, 0.1]; inds["gauss"][:, :, 0] = True
x["sine"][:, :, 0] = [1.2, 7.3, 1.0] + 1e-2 * rs.randn(T, W, 3); inds["sine"][:, :, 0] = True
eng.upload(x, inds, betas=make_ladder(6, ntemps=T)); eng.eval_state()
eng.set_mh_scale(np.full((2, 3), 1e-2) * [[1, 1, 0.1], [1, 1, 1]])
for n in (1, 70, 130):                       # (calls that end on, before and after a template refresh)
    eng.step(n)
eng.synchronize()
x1, inds1, L1, P1, betas1 = eng.download()
c = eng.counters()
np.savez(sys.argv[2], xg=x1["gauss"], xs=x1["sine"], ig=inds1["gauss"], js=inds1["sine"], L=L1, P=P1, betas=betas1,
         acc_mh=c["accepted_mh"], acc_bd=c["accepted_bd"], swaps_total=c["swaps_total"], swaps_last=c["swaps_last"])
eng.close()
"""


def test_rj_adaptation_folded_into_the_next_launch_changes_nothing(tmp_path):
    """hens_rj_step runs the ladder adaptation behind a cascade INSIDE the next k_rj launch (wave 0 adapts and publishes, the others
    pick up their rung's beta at their accept test); HENS_NO_FOLD=1 keeps k_adapt as a launch of its own.  Same arithmetic, same
    order: 201 iterations of both must agree in every bit - coordinates, masks, log-probabilities, the ladder and every counter."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({}, {"HENS_NO_FOLD": "1"}):
        out = str(tmp_path / f"{len(outs)}.npz")
        e = dict(os.environ, **env)
        if not env:
            e.pop("HENS_NO_FOLD", None)
        r = subprocess.run([sys.executable, "-c", _FOLD_WORKER, root, out], env=e, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(dict(np.load(out)))
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), f"folded adaptation vs k_adapt: `{k}` differs"
    assert outs[0]["acc_mh"].sum() > 0 and outs[0]["acc_bd"].sum() > 0 and outs[0]["swaps_total"].sum() > 0
    assert not np.array_equal(outs[0]["betas"], __import__("eryn_amd.moves.tempering", fromlist=["make_ladder"]).make_ladder(6, ntemps=6)), "the ladder must have moved"


def test_pulse_centre_far_outside_the_data_grid_is_not_a_nan():
    """the production likelihood's pulse recurrence e_{k+1} = e_k r_k formed r_0 = exp(-((2 dx) h + h^2) / (2 c^2)) without a
    bound - a centre 1 000 grid steps past the data with c close to the grid step makes the exponent +790: r_0 = inf while e_0 has
    underflowed to 0, 0 x inf = NaN, and hens_rj_step aborted with "likelihood is returning Nan".  The prior box below lets centres
    sit up to 3 units (750 steps) outside; the chain must run and its log-likelihoods must be the direct formula's."""
    from eryn_amd.rj import RJEngine, TemplateBranch
    from oracle import eryn_oracle_rj as orj
    T, W, N = 2, 64, 500
    t = np.linspace(-1, 1, N)                                              # h = 0.004
    rs = np.random.RandomState(5)
    y = 1.0 * np.sin(2 * np.pi * 3.0 * t + 0.5) + 0.5 * rs.randn(N)
    boxes = {"gauss": [(2.5, 3.5), (-1.0, 4.0), (0.0041, 0.0060)], "sine": [(0.5, 1.5), (1.0, 20.0), (0.0, 2 * np.pi)]}
    brs = [TemplateBranch("gauss", "pulse", boxes["gauss"], 3, 0), TemplateBranch("sine", "sine", boxes["sine"], 2, 0)]
    eng = RJEngine(T, W, brs, t, y, 0.5, seed=3)
    x = {"gauss": np.zeros((T, W, 3, 3)), "sine": np.zeros((T, W, 2, 3))}
    inds = {"gauss": np.zeros((T, W, 3), dtype=bool), "sine": np.zeros((T, W, 2), dtype=bool)}
    x["gauss"][:, :, 0] = [3.0, 3.0, 0.0045]                                # dx h / c^2 = 4 x 0.004 / 2.0e-5 = 790 at the first point
    x["gauss"][:, :, 1] = [3.0, 1.9, 0.0042]
    x["gauss"][:, :, :2, 1] += 0.05 * rs.rand(T, W, 2)
    x["sine"][:, :, 0] = [1.0, 3.0, 0.5]
    inds["gauss"][:, :, :2] = True
    inds["sine"][:, :, 0] = True
    eng.upload(x, inds, betas=np.array([1.0, 0.5]))
    eng.eval_state()
    eng.set_mh_scale(np.array([[1e-2, 1e-2, 1e-5], [1e-2, 1e-2, 1e-2]]))
    eng.step(6)                                                             # (raised RuntimeError before the clamp)
    eng.synchronize()
    x1, inds1, L1, P1, _ = eng.download()
    okind = {"pulse": orj.KIND_PULSE, "sine": orj.KIND_SINE}
    obr = [orj.Branch("gauss", okind["pulse"], boxes["gauss"], 3, 0), orj.Branch("sine", okind["sine"], boxes["sine"], 2, 0)]
    ref = orj.template_log_like(x1, inds1, obr, t, y, 0.5)
    assert np.isfinite(L1).all()
    assert (x1["gauss"][:, :, :, 1][inds1["gauss"]] > 1.5).any(), "the far-out centres must still be there"
    tol.check_logl(L1, ref, RTOL_L, "template log-like with pulse centres far outside the grid")
    eng.close()
