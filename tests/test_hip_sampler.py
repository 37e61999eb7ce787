"""GPU tests of the host-side mirror: eryn_amd.ensemble.EnsembleSampler driving the HIP move exactly
the way the reference's sampler drives its StretchMove (same seeds -> same chain), read like the
reference's own tests (tests/test_eryn.py::test_base / test_pt) but WITH value assertions against the
fixtures captured from the reference."""
import os

import numpy as np
import pytest

from eryn_amd.ensemble import EnsembleSampler
from eryn_amd.likelihood import GaussianLikelihood
from eryn_amd.prior import ProbDistContainer, uniform_dist
from eryn_amd.state import State

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["f2_pt", "f3_oddW", "f4_narrowbox", "f5_noadapt", "f5_nopermute", "f1_plumbing", "f8_nsplits3"])
def test_dropin_sampler_reproduces_reference_chain(name, golden_dir):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    T, W, D, n = int(fx["T"]), int(fx["W"]), int(fx["D"]), int(fx["nsteps"])
    box = float(fx["box"])
    np.random.seed(int(fx["seed_construct"]))                 # R := snapshot of G at construction
    priors = ProbDistContainer({i: uniform_dist(-box, box) for i in range(D)})
    kw = {}
    if "betas0" in fx.files:
        kw["tempering_kwargs"] = dict(ntemps=T, adaptive=bool(fx["adaptive"]), permute=bool(fx["permute"]))
    if "nsplits" in fx.files and int(fx["nsplits"]) != 2:     # RedBlueMove(nsplits=...): three sets (red_blue.py:41-47,148)
        from eryn_amd.moves import StretchMove
        kw["moves"] = StretchMove(nsplits=int(fx["nsplits"]))
    s = EnsembleSampler(W, D, GaussianLikelihood(fx["mu"], fx["invcov"]), priors, **kw)
    np.random.seed(int(fx["seed_run"]))                       # G for the run
    it = 0
    for state in s.sample(fx["x0"], iterations=n, store=True):
        pre = f"it{it}_"
        x = state.branches["model_0"].coords[:, :, 0, :]
        assert np.array_equal(x, fx[pre + "x"]), f"positions differ at iteration {it}"
        assert np.array_equal(state.log_prior, fx[pre + "P"])
        np.testing.assert_allclose(state.log_like, fx[pre + "L"], rtol=1e-11, atol=0)
        if pre + "betas" in fx.files:
            np.testing.assert_allclose(state.betas, fx[pre + "betas"], rtol=1e-12, atol=0)
            assert np.array_equal(s.temperature_control.swaps_accepted, fx[pre + "swaps_accepted"])
        it += 1
    assert np.array_equal(s.moves[0].accepted, fx["accepted_total"])
    assert s.moves[0].num_proposals == int(fx["num_proposals"])
    assert s.backend.iteration == n and s.get_chain()["model_0"].shape == (n, T, W, 1, D)


def test_philox_sampler_targets_the_gaussian():
    """Production RNG: the cold chain must sample the analytic Gaussian (statistical check the
    reference's smoke tests never made)."""
    T, W, D = 4, 2048, 8
    rs = np.random.RandomState(0)
    A = rs.randn(D, D)
    mu = 0.1 * rs.randn(D)
    cov = A @ A.T / D + np.eye(D)
    priors = {i: uniform_dist(-50.0, 50.0) for i in range(D)}
    s = EnsembleSampler(W, D, GaussianLikelihood(mu, np.linalg.inv(cov)), priors,
                        tempering_kwargs=dict(ntemps=T), rng="philox", seed=11)
    x0 = np.random.RandomState(1).randn(T, W, D)
    state = s.run_mcmc(x0, 20, burn=300, thin_by=10)
    chain = s.get_chain()["model_0"][:, 0, :, 0, :].reshape(-1, D)      # cold rung, 20 x 2048 samples
    assert np.abs(chain.mean(0) - mu).max() < 0.05
    assert np.linalg.norm(np.cov(chain.T) - cov) / np.linalg.norm(cov) < 0.05
    assert abs(state.log_like[0].mean() + D / 2) < 0.3
    acc = s.moves[0].acceptance_fraction
    assert 0.2 < acc[0].mean() < 0.8
    assert np.all(np.diff(state.betas) < 0) and state.betas[0] == 1.0
    assert s.temperature_control.time == 300 + 200


@pytest.mark.parametrize("T,W,D,periodic", [(10, 1024, 11, None), (5, 2048, 16, None), (6, 1024, 8, {2: 40.0})])
def test_philox_sampler_targets_the_gaussian_on_padded_rows_and_short_tiles(T, W, D, periodic):
    """The same statistical check on what the end of round 2 added to the production path: ladders that do not divide 128
    (short tiles, one launch per iteration), rows padded from D = 11 to 16, and a periodic parameter whose period is far
    wider than the target (wrapping must not disturb the interior of the distribution).  Every rung's mean log-likelihood
    is that of the tempered Gaussian, -D / 2 / max(beta, ...) -> L beta = -D / 2."""
    rs = np.random.RandomState(0)
    A = rs.randn(D, D)
    mu = 0.1 * rs.randn(D)
    if periodic:
        mu[list(periodic)] = 20.0                                       # mid-period: the wrap is far away
    cov = A @ A.T / D + np.eye(D)
    priors = {i: uniform_dist(-50.0, 50.0) for i in range(D)}
    kw = dict(periodic={"model_0": periodic}) if periodic else {}
    s = EnsembleSampler(W, D, GaussianLikelihood(mu, np.linalg.inv(cov)), priors,
                        tempering_kwargs=dict(ntemps=T), rng="philox", seed=23, **kw)
    x0 = mu + np.random.RandomState(1).randn(T, W, D)
    state = s.run_mcmc(x0, 20, burn=400, thin_by=10)
    chain = s.get_chain()["model_0"][:, 0, :, 0, :].reshape(-1, D)
    assert np.abs(chain.mean(0) - mu).max() < 0.06
    assert np.linalg.norm(np.cov(chain.T) - cov) / np.linalg.norm(cov) < 0.06
    tempered = (state.log_like * state.betas[:, None]).mean(axis=1)      # E[beta L] = -D / 2 on every rung
    assert np.all(np.abs(tempered[state.betas > 0.05] + D / 2) < 0.12 * D / 2 + 0.3)
    assert np.all(np.diff(state.betas) < 0) and state.betas[0] == 1.0
    assert 0.15 < s.moves[0].acceptance_fraction[0].mean() < 0.8


def test_state_round_trip_and_errors():
    D, W = 4, 16
    like = GaussianLikelihood(np.zeros(D), np.eye(D))
    priors = {i: uniform_dist(-1.0, 1.0) for i in range(D)}
    s = EnsembleSampler(W, D, like, priors)
    x_bad = np.full((1, W, D), 2.0)                            # outside the box -> -inf prior
    with pytest.raises(ValueError, match="log_prior"):
        next(s.sample(x_bad, iterations=1))
    with pytest.raises(ValueError):
        next(s.sample(np.zeros((2, W, D)), iterations=1))      # wrong ntemps
    x0 = np.random.RandomState(0).uniform(-0.5, 0.5, size=(1, W, D))
    st = s.run_mcmc(State(x0), 3)
    assert st.branches["model_0"].coords.shape == (1, W, 1, D)
    lp = s.compute_log_prior({"model_0": st.branches["model_0"].coords})
    assert np.array_equal(lp, st.log_prior)


@pytest.mark.parametrize("T,W,D,like", [(1, 100, 3, "dense"), (3, 130, 7, "dense"), (2, 64, 16, "diag"),
                                        (4, 256, 64, "dense"), (64, 80, 8, "dense"), (5, 333, 32, "dense")])
def test_philox_step_shapes(T, W, D, like):
    """hens_step on awkward shapes: generic row widths, odd walker counts, no tempering, many rungs
    (T = 64 uses the stand-alone adaptation kernel's limit case)."""
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.moves.tempering import make_ladder
    rs = np.random.RandomState(0)
    A = rs.randn(D, D)
    mu = 0.1 * rs.randn(D)
    cov = A @ A.T / D + np.eye(D)
    prec = np.linalg.inv(cov) if like == "dense" else 1.0 / np.diag(cov)
    eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, prec), -20.0, 20.0, seed=3)
    eng.upload(np.random.RandomState(1).randn(T, W, D), betas=make_ladder(D, ntemps=T) if T > 1 else None)
    eng.eval_state()
    x0, L0, P0, _ = eng.download()
    ref = -0.5 * np.einsum("twi,ij,twj->tw", x0 - mu, prec if like == "dense" else np.diag(prec), x0 - mu)
    np.testing.assert_allclose(L0, ref, rtol=1e-11)
    n = 60
    eng.step(n)
    x, L, P, betas = eng.download()
    c = eng.counters()
    assert np.isfinite(x).all() and np.abs(x).max() <= 20.0
    np.testing.assert_allclose(L, -0.5 * np.einsum("twi,ij,twj->tw", x - mu, prec if like == "dense" else np.diag(prec), x - mu), rtol=1e-10)
    assert np.all(P == P0[0, 0])
    assert c["num_proposals"] == n and c["accepted"].shape == (T, W)
    acc = c["accepted"].mean() / n
    assert 0.03 < acc < 0.97, acc
    if T > 1:
        assert c["adapt_time"] == n and betas[0] == 1.0 and np.all(np.diff(betas) < 0)
        assert np.all(c["swaps_total"] <= n * W) and c["swaps_total"].sum() > 0
    eng.close()


def _ll_vec(x, mu, invcov):                     # tests/test_eryn.py:33-35, batched
    diff = x - mu
    return -0.5 * (diff * np.dot(invcov, diff.T).T).sum(axis=1)


def _ll_single(x, mu, invcov):                  # tests/test_eryn.py:33-35
    diff = x - mu
    return -0.5 * (diff * np.dot(invcov, diff.T).T).sum()


@pytest.mark.parametrize("name", ["f1_plumbing", "f2_pt", "f4_narrowbox", "f7_tmaxinf"])
def test_python_callable_likelihood_reproduces_reference_chain(name, golden_dir):
    """an arbitrary Python log_like_fn (here the reference tests' own function, vectorised
    and per-walker) with proposal / prior / accept / update / PT on the device.  The likelihood values
    come from the same NumPy code as the reference's, so the whole chain is bit-identical."""
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    T, W, D, n = int(fx["T"]), int(fx["W"]), int(fx["D"]), int(fx["nsteps"])
    box = float(fx["box"])
    vec = bool(fx["vectorize"])
    np.random.seed(int(fx["seed_construct"]))
    priors = ProbDistContainer({i: uniform_dist(-box, box) for i in range(D)})
    kw = {}
    if "betas0" in fx.files:
        kw["tempering_kwargs"] = dict(betas=fx["betas0"].copy(), adaptive=bool(fx["adaptive"]), permute=bool(fx["permute"]))
    s = EnsembleSampler(W, D, _ll_vec if vec else _ll_single, priors, args=[fx["mu"], fx["invcov"]], vectorize=vec, **kw)
    np.random.seed(int(fx["seed_run"]))
    it = 0
    for state in s.sample(fx["x0"], iterations=n, store=False):
        pre = f"it{it}_"
        assert np.array_equal(state.branches["model_0"].coords[:, :, 0, :], fx[pre + "x"]), f"iteration {it}"
        assert np.array_equal(state.log_like, fx[pre + "L"]), f"log-like differs at iteration {it}"
        assert np.array_equal(state.log_prior, fx[pre + "P"])
        if pre + "betas" in fx.files:
            np.testing.assert_allclose(state.betas, fx[pre + "betas"], rtol=1e-12, atol=0)
        it += 1
    assert np.array_equal(s.moves[0].accepted, fx["accepted_total"])


def test_num_repeats_in_model_and_thinning_reproduce_reference_chain(golden_dir):
    """ensemble.py:243-256, 963-1045: num_repeats_in_model = 3 proposals per sub-iteration, thin_by = 2 sub-iterations per
    stored step - the fixture was captured from the reference's own sampler (tests/golden/make_golden_repeats.py)."""
    fx = np.load(os.path.join(golden_dir, "r1_repeats3_thin2.npz"))
    T, W, D, n = int(fx["T"]), int(fx["W"]), int(fx["D"]), int(fx["nsteps"])
    box = float(fx["box"])
    np.random.seed(int(fx["seed_construct"]))
    priors = ProbDistContainer({i: uniform_dist(-box, box) for i in range(D)})
    tuned = []

    s = EnsembleSampler(W, D, GaussianLikelihood(fx["mu"], fx["invcov"]), priors, tempering_kwargs=dict(ntemps=T),
                        num_repeats_in_model=int(fx["repeats"]))
    s.moves[0].tune = lambda state, accepted: tuned.append(accepted.copy())       # the reference's hook (ensemble.py:983-984)
    np.random.seed(int(fx["seed_run"]))
    it = 0
    for state in s.sample(fx["x0"], iterations=n, thin_by=int(fx["thin_by"]), store=True, tune=True):
        pre = f"it{it}_"
        assert np.array_equal(state.branches["model_0"].coords[:, :, 0, :], fx[pre + "x"]), f"positions differ at stored step {it}"
        assert np.array_equal(state.log_prior, fx[pre + "P"])
        np.testing.assert_allclose(state.log_like, fx[pre + "L"], rtol=1e-11, atol=0)
        np.testing.assert_allclose(state.betas, fx[pre + "betas"], rtol=1e-12, atol=0)
        assert np.array_equal(s.temperature_control.swaps_accepted, fx[pre + "swaps_accepted"])
        assert np.array_equal(s.backend.accepted, fx[pre + "backend_accepted"])
        assert np.array_equal(s.backend.swaps_accepted, fx[pre + "backend_swaps"])
        it += 1
    assert np.array_equal(s.moves[0].accepted, fx["move_accepted"]) and s.moves[0].num_proposals == int(fx["num_proposals"])
    assert len(tuned) == int(fx["num_proposals"]) and np.array_equal(sum(tuned), fx["move_accepted"])
    assert np.array_equal(s.get_chain()["model_0"][:, :, :, 0, :], fx["chain"])


@pytest.mark.parametrize("T,W,D,reps,thin", [(4, 256, 8, 1, 1), (8, 512, 32, 3, 2), (1, 64, 8, 1, 3)])
def test_philox_chain_resumes_bit_identically_from_a_stored_state(T, W, D, reps, thin):
    """The device-side form of the reference's random_state checkpoint (backends/backend.py:1014-1091, ensemble.py:605-647):
    a stored State carries (seed, Philox iteration counter, adaptation time); a NEW sampler object with the same seed continues
    the chain from it bit for bit (hens_set_iteration)."""
    rs = np.random.RandomState(3)
    A = rs.randn(D, D)
    mu, invcov = 0.1 * rs.randn(D), np.linalg.inv(A @ A.T / D + np.eye(D))
    priors = {i: uniform_dist(-20.0, 20.0) for i in range(D)}
    x0 = np.random.RandomState(1).randn(T, W, D)

    def sampler():
        kw = dict(tempering_kwargs=dict(ntemps=T)) if T > 1 else {}
        return EnsembleSampler(W, D, GaussianLikelihood(mu, invcov), priors, rng="philox", seed=77,
                               num_repeats_in_model=reps, **kw)

    a = sampler()
    whole = [State(st, copy=True) for st in a.sample(x0 if T > 1 else x0[0], iterations=9, thin_by=thin, store=True)]
    b = sampler()
    first = [State(st, copy=True) for st in b.sample(x0 if T > 1 else x0[0], iterations=4, thin_by=thin, store=True)]
    assert first[-1].random_state == whole[3].random_state and first[-1].random_state[0] == "philox"
    assert first[-1].random_state[2] == 4 * thin * reps
    c = sampler()                                                       # a new context: nothing but the stored State travels
    rest = [State(st, copy=True) for st in c.sample(first[-1], iterations=5, thin_by=thin, store=True)]
    for k, (u, v) in enumerate(zip(whole, first + rest)):
        for f in ("log_like", "log_prior", "betas"):
            fu, fv = getattr(u, f), getattr(v, f)
            assert (fu is None and fv is None) or np.array_equal(fu, fv), f"{f} differs at stored step {k}"
        assert np.array_equal(u.branches["model_0"].coords, v.branches["model_0"].coords), f"positions differ at stored step {k}"
        assert u.random_state == v.random_state
    with pytest.raises(ValueError):                                     # another seed is another stream
        EnsembleSampler(W, D, GaussianLikelihood(mu, invcov), priors, rng="philox", seed=78, num_repeats_in_model=reps,
                        **(dict(tempering_kwargs=dict(ntemps=T)) if T > 1 else {})).run_mcmc(first[-1], 1)


@pytest.mark.parametrize("mix", [False, True])
def test_philox_thinned_accept_mask_is_the_last_sub_iterations(mix):
    """ensemble.py:968-979: with thin_by > 1 the stored accept mask is the LAST sub-iteration's.  The thinned run keeps the
    counters in front of that sub-iteration on the device (hens_step_marked); an unthinned run of the same chain gives the
    per-iteration masks to compare with."""
    from eryn_amd.moves import GaussianMove, StretchMove
    T, W, D, thin, reps, n = 4, 256, 8, 3, 2, 5
    rs = np.random.RandomState(5)
    A = rs.randn(D, D)
    mu, invcov = 0.1 * rs.randn(D), np.linalg.inv(A @ A.T / D + np.eye(D))
    priors = {i: uniform_dist(-20.0, 20.0) for i in range(D)}
    x0 = np.random.RandomState(2).randn(T, W, D)

    def sampler():
        moves = [(StretchMove(), 0.5), (GaussianMove({"model_0": 0.05 * np.eye(D)}), 0.5)] if mix else None
        return EnsembleSampler(W, D, GaussianLikelihood(mu, invcov), priors, rng="philox", seed=11, moves=moves,
                               num_repeats_in_model=reps, tempering_kwargs=dict(ntemps=T))

    a, masks, last = sampler(), [], None
    for _ in a.sample(x0, iterations=n * thin, thin_by=1, store=True):
        cur = a.backend.accepted.copy()
        masks.append(cur if last is None else cur - last)
        last = cur
    b, got, last = sampler(), [], None
    for _ in b.sample(x0, iterations=n, thin_by=thin, store=True):
        cur = b.backend.accepted.copy()
        got.append(cur if last is None else cur - last)
        last = cur
    for k in range(n):
        assert np.array_equal(got[k], masks[k * thin + thin - 1]), f"stored step {k}"
    assert any(m.any() for m in got)
    assert np.array_equal(a.get_chain()["model_0"][thin - 1::thin], b.get_chain()["model_0"])


def test_philox_tune_hook_is_called_per_proposal_with_that_moves_own_mask():
    """ensemble.py:969-984: `move.tune(state, accepted_out)` after EVERY proposal, with that proposal's mask - also with
    rng="philox", thin_by > 1, num_repeats_in_model > 1 and two moves in the mix; tuning changes nothing about the chain."""
    from eryn_amd.moves import GaussianMove, StretchMove
    T, W, D, thin, reps, n = 4, 256, 8, 2, 3, 4
    rs = np.random.RandomState(5)
    A = rs.randn(D, D)
    mu, invcov = 0.1 * rs.randn(D), np.linalg.inv(A @ A.T / D + np.eye(D))
    priors = {i: uniform_dist(-20.0, 20.0) for i in range(D)}
    x0 = np.random.RandomState(2).randn(T, W, D)

    def sampler():
        moves = [(StretchMove(), 0.5), (GaussianMove({"model_0": 0.05 * np.eye(D)}), 0.5)]
        return EnsembleSampler(W, D, GaussianLikelihood(mu, invcov), priors, rng="philox", seed=11, moves=moves,
                               num_repeats_in_model=reps, tempering_kwargs=dict(ntemps=T))

    a = sampler()
    plain = [State(st, copy=True) for st in a.sample(x0, iterations=n, thin_by=thin, store=True)]
    b = sampler()
    calls = {0: [], 1: []}
    for k, m in enumerate(b.moves):
        m.tune = (lambda kk: (lambda state, accepted: calls[kk].append((accepted.copy(), state.log_like.copy()))))(k)
    tuned = [State(st, copy=True) for st in b.sample(x0, iterations=n, thin_by=thin, store=True, tune=True)]
    assert len(calls[0]) + len(calls[1]) == n * thin * reps and len(calls[0]) > 0 and len(calls[1]) > 0
    for k, m in enumerate(b.moves):                      # every move saw exactly its own proposals' masks
        assert len(calls[k]) == m.num_proposals
        assert np.array_equal(sum(c[0] for c in calls[k]), m.accepted)
    for u, v in zip(plain, tuned):                       # the hook is an observer: same chain, same stored masks
        assert np.array_equal(u.branches["model_0"].coords, v.branches["model_0"].coords) and np.array_equal(u.log_like, v.log_like)
    assert np.array_equal(a.backend.accepted, b.backend.accepted)


def test_philox_tune_hook_that_rescales_the_gaussian_move_reaches_the_device():
    """A `tune` hook mutates the move object and the NEXT proposal reads it (ensemble.py:983-984; the reference's shipped tuner does
    so with `move.a`).  For a GaussianMove that means the retuned scale must be handed to the device: a chain whose hook shrinks
    the scale after the second proposal equals, from there on, a chain resumed from that state by a sampler built with the small
    scale (same seed: the Philox checkpoint in the stored State carries the stream across)."""
    from eryn_amd.moves import GaussianMove
    T, W, D, n = 4, 256, 8, 5
    rs = np.random.RandomState(5)
    A = rs.randn(D, D)
    mu, invcov = 0.1 * rs.randn(D), np.linalg.inv(A @ A.T / D + np.eye(D))
    priors = {i: uniform_dist(-20.0, 20.0) for i in range(D)}
    x0 = np.random.RandomState(2).randn(T, W, D)

    def sampler(cov):
        return EnsembleSampler(W, D, GaussianLikelihood(mu, invcov), priors, rng="philox", seed=11, moves=GaussianMove({"model_0": cov}),
                               tempering_kwargs=dict(ntemps=T))

    a = sampler(0.3)
    ncall = [0]

    def hook(state, accepted):
        ncall[0] += 1
        if ncall[0] == 2:
            a.moves[0].scale = np.sqrt(0.01)                 # (what GaussianMove({"model_0": 0.01}) holds: gaussian.py:44-46)
    a.moves[0].tune = hook
    tuned = [State(st, copy=True) for st in a.sample(x0, iterations=n, store=True, tune=True)]
    b = sampler(0.01)
    rest = [State(st, copy=True) for st in b.sample(tuned[1], iterations=n - 2, store=True)]
    for k, (u, v) in enumerate(zip(tuned[2:], rest)):
        assert np.array_equal(u.branches["model_0"].coords, v.branches["model_0"].coords), f"positions differ {k + 1} steps after the retune"
        assert np.array_equal(u.log_like, v.log_like) and u.random_state == v.random_state
    c = sampler(0.3)                                         # and the retune did change the chain
    plain = [State(st, copy=True) for st in c.sample(x0, iterations=n, store=True)]
    assert np.array_equal(plain[1].branches["model_0"].coords, tuned[1].branches["model_0"].coords)
    assert not np.array_equal(plain[2].branches["model_0"].coords, tuned[2].branches["model_0"].coords)


def test_philox_sampler_steps_a_three_set_stretch_move():
    """EnsembleSampler(rng="philox") with StretchMove(nsplits=3) (red_blue.py:41-47): the device draws the three sets itself
    (tests/test_hip_replay.py holds that path to the oracle); here the sampler plumbing - counters, stored chain, thinning."""
    from eryn_amd.moves import StretchMove
    T, W, D = 4, 300, 8
    rs = np.random.RandomState(5)
    A = rs.randn(D, D)
    mu, invcov = 0.1 * rs.randn(D), np.linalg.inv(A @ A.T / D + np.eye(D))
    priors = {i: uniform_dist(-20.0, 20.0) for i in range(D)}
    s = EnsembleSampler(W, D, GaussianLikelihood(mu, invcov), priors, rng="philox", seed=3, moves=StretchMove(nsplits=3),
                        tempering_kwargs=dict(ntemps=T))
    last = s.run_mcmc(np.random.RandomState(2).randn(T, W, D), 6, thin_by=2)
    assert s.moves[0].num_proposals == 12 and 0.05 < s.moves[0].accepted.mean() / 12 < 0.9
    assert np.isfinite(last.log_like).all() and s.get_chain()["model_0"].shape[0] == 6


# ---- round 6: device-draw drop-in moves with a device-resident State ---------------------------------------------
def _gauss(D, seed=3):
    rs = np.random.RandomState(seed)
    A = rs.randn(D, D)
    return 0.1 * rs.randn(D), np.linalg.inv(A @ A.T / D + np.eye(D))


@pytest.mark.parametrize("T,W,D,thin", [(8, 512, 32, 1), (4, 256, 16, 5), (16, 1024, 64, 3)])
def test_device_draw_move_in_the_numpy_loop_is_hens_step(T, W, D, thin):
    """``StretchMove(rng="philox")`` under eryn_amd's host loop (rng="numpy": one ``propose()`` per iteration, as the reference's
    sampler calls it) against ONE ``hens_step`` call of the same length on a context of its own: same seed, same state - positions,
    log-probabilities, ladder bit for bit, accept counters equal to the summed masks.  The walkers cross the boundary at stored
    steps only (DeviceState)."""
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.moves import StretchMove
    from eryn_amd.moves.tempering import make_ladder
    from eryn_amd.state import DeviceState
    mu, invcov = _gauss(D)
    like = GaussianLikelihood(mu, invcov)
    priors = {i: uniform_dist(-20.0, 20.0) for i in range(D)}
    x0 = np.random.RandomState(1).randn(T, W, D)
    n = 6
    s = EnsembleSampler(W, D, like, priors, tempering_kwargs=dict(ntemps=T), moves=StretchMove(rng="philox"), seed=17)
    eng = s.engine
    states = []
    for st in s.sample(x0, iterations=n, thin_by=thin, store=True):
        assert isinstance(st, DeviceState)
        states.append(st)
    assert getattr(eng, "lazy_downloads", 0) == n, "the walkers must be copied back at stored steps only"
    ref = HipEnsemble(T, W, D, like, -20.0, 20.0, seed=17)
    ref.upload(x0, betas=make_ladder(D, ntemps=T))
    ref.eval_state()
    for k in range(n):
        ref.step(thin)
        x, L, P, betas = ref.download()
        st = states[k]
        assert np.array_equal(st.branches["model_0"].coords[:, :, 0, :], x), f"positions at stored step {k}"
        assert np.array_equal(st.log_like, L) and np.array_equal(st.log_prior, P) and np.array_equal(st.betas, betas)
    c = ref.counters()
    assert np.array_equal(s.moves[0].accepted, c["accepted"]) and s.moves[0].num_proposals == n * thin
    assert np.array_equal(s.temperature_control.swaps_accepted, c["swaps_last"])
    assert np.array_equal(s.get_chain()["model_0"][-1][:, :, 0, :], x)
    ref.close()


def test_device_draw_move_mix_in_the_numpy_loop():
    """StretchMove + GaussianMove, both rng="philox", chosen per proposal by the sampler's own stream (ensemble.py:971): every
    proposal is one device iteration of ITS move; the accept counters of the two moves partition the proposals."""
    from eryn_amd.moves import GaussianMove, StretchMove
    T, W, D = 4, 512, 32
    mu, invcov = _gauss(D)
    like = GaussianLikelihood(mu, invcov)
    priors = {i: uniform_dist(-20.0, 20.0) for i in range(D)}
    mvs = [(StretchMove(rng="philox"), 0.6), (GaussianMove({"model_0": 0.01}, rng="philox"), 0.4)]
    np.random.seed(5)
    s = EnsembleSampler(W, D, like, priors, tempering_kwargs=dict(ntemps=T), moves=mvs, seed=3)
    last = s.run_mcmc(np.random.RandomState(1).randn(T, W, D), 40, store=False)
    st, g = s.moves
    assert st.num_proposals + g.num_proposals == 40 and st.num_proposals > 5 and g.num_proposals > 5
    c, cm = s.engine.counters(), s.engine.mh_counters()
    assert np.array_equal(st.accepted, c["accepted"]) and np.array_equal(g.accepted, cm["accepted"])
    assert c["num_proposals"] == st.num_proposals and cm["num_proposals"] == g.num_proposals
    assert np.isfinite(last.log_like).all() and st.accepted.sum() > 0 and g.accepted.sum() > 0


def test_numpy_loop_with_the_lazy_device_move_runs_config_2_under_150_us_per_iteration():
    """the drop-in move under a host loop that calls ``propose()`` once per iteration ran at ~1 200 us per iteration
    (17.8 MB down + up per proposal).  With the device-draw move and the device-resident State the per-proposal traffic is the
    accept mask (64 kB) + swap counts + ladder: <= 150 us per iteration at BASELINE config 2 (16 x 4096 x 32), Python loop included."""
    import time
    from eryn_amd.moves import StretchMove
    T, W, D = 16, 4096, 32
    rs = np.random.RandomState(0)
    A = rs.randn(D, D)
    mu, invcov = 0.1 * rs.randn(D), np.linalg.inv(A @ A.T / D + np.eye(D))
    priors = {i: uniform_dist(-50.0, 50.0) for i in range(D)}
    s = EnsembleSampler(W, D, GaussianLikelihood(mu, invcov), priors, tempering_kwargs=dict(ntemps=T), moves=StretchMove(rng="philox"), seed=2024)
    st = s.run_mcmc(np.random.RandomState(1).randn(T, W, D), 100, store=False)      # warm-up
    best = np.inf
    for _ in range(3):
        t0 = time.perf_counter()
        st = s.run_mcmc(st, 500, store=False)
        best = min(best, (time.perf_counter() - t0) / 500)
    print(f"numpy loop + lazy device move: {best * 1e6:.1f} us per iteration at config 2")
    assert best * 1e6 <= 150.0, f"{best * 1e6:.1f} us per iteration"
    assert s.engine.lazy_downloads <= 8
    assert np.isfinite(st.log_like).all()
