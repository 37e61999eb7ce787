"""GPU tests of the Metropolis-Hastings path (GaussianMove / MHMove on the device, weighted
move mix).  The fixtures come from the real reference (tests/golden/make_golden_mh.py); the oracle that
interprets them is pinned on the same files in tests/test_oracle_golden.py."""
import os

import numpy as np
import pytest

from eryn_amd.ensemble import EnsembleSampler
from eryn_amd.likelihood import GaussianLikelihood
from eryn_amd.moves import GaussianMove, StretchMove
from eryn_amd.prior import ProbDistContainer, uniform_dist
from tests import tolerance_log as tol
from tests.test_oracle_golden import MH_FIXTURES, build_mh_oracle, mh_moves_from_fixture

pytestmark = pytest.mark.gpu


def _sampler_from_fixture(fx, **kw):
    T, W, D = int(fx["T"]), int(fx["W"]), int(fx["D"])
    box = float(fx["box"])
    specs = mh_moves_from_fixture(fx, lambda cov, mode, factor: ("gauss", cov, mode, factor))
    moves = []
    for m, w in specs:
        if m == "stretch":
            moves.append((StretchMove(a=2.0), w))
        else:
            _, cov, mode, factor = m
            moves.append((GaussianMove({"model_0": cov if cov.ndim else float(cov)}, mode=mode, factor=factor), w))
    np.random.seed(int(fx["seed_construct"]))                 # R := snapshot of G at construction
    priors = ProbDistContainer({i: uniform_dist(-box, box) for i in range(D)})
    if "betas0" in fx.files:
        kw["tempering_kwargs"] = dict(ntemps=T)
    if "period" in fx.files:                                   # ensemble.py:165-168
        kw["periodic"] = {"model_0": {int(d): float(p) for d, p in enumerate(fx["period"]) if p > 0}}
    return EnsembleSampler(W, D, GaussianLikelihood(fx["mu"], fx["invcov"]), priors, moves=moves, **kw)


@pytest.mark.parametrize("name", MH_FIXTURES)
def test_dropin_sampler_reproduces_reference_chain_with_mh_moves(name, golden_dir):
    """Same seeds -> the reference's chain: move choice, proposals, accept masks, PT, adapted ladder."""
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    n = int(fx["nsteps"])
    s = _sampler_from_fixture(fx)
    np.random.seed(int(fx["seed_run"]))
    it = 0
    for state in s.sample(fx["x0"], iterations=n, store=False):
        pre = f"it{it}_"
        x = state.branches["model_0"].coords[:, :, 0, :]
        assert np.array_equal(x, fx[pre + "x"]), f"positions differ at iteration {it}"
        assert np.array_equal(state.log_prior, fx[pre + "P"])
        np.testing.assert_allclose(state.log_like, fx[pre + "L"], rtol=1e-11, atol=0)
        if pre + "betas" in fx.files:
            np.testing.assert_allclose(state.betas, fx[pre + "betas"], rtol=1e-12, atol=0)
            assert np.array_equal(s.temperature_control.swaps_accepted, fx[pre + "swaps_accepted"])
        it += 1
    for i, m in enumerate(s.moves):
        assert m.num_proposals == int(fx[f"move{i}_num_proposals"])
        assert np.array_equal(m.accepted, fx[f"move{i}_accepted"])


@pytest.mark.parametrize("name", ["p1_stretch_periodic", "p2_mix_periodic", "p3_stretch_periodic_untempered"])
def test_stretch_split_teacher_forced_with_periodic_parameters(name, golden_dir):
    """hens_stretch_split with hens_set_periodic, one half-step at a time from the oracle's state, against the oracle's
    intermediates (which reproduce the reference's chain on these fixtures, tests/test_oracle_golden.py): distances the
    short way round and wrapped proposals (stretch.py:136-154) -> accept masks and positions bit for bit."""
    from eryn_amd.engine import HipEnsemble
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    T, W, D, box = int(fx["T"]), int(fx["W"]), int(fx["D"]), float(fx["box"])
    o = build_mh_oracle(fx)
    eng = HipEnsemble(T, W, D, GaussianLikelihood(fx["mu"], fx["invcov"]), -box, box, tempered=o.tempered)
    eng.set_periodic(fx["period"])
    nst = 0
    for it in range(int(fx["nsteps"])):
        prev = (o.x.copy(), o.L.copy(), o.P.copy(), None if o.betas is None else o.betas.copy())
        o.iteration()
        rec = o.trace[-1]
        if "labels" not in rec:
            continue
        nst += 1
        eng.upload(*prev)
        for sp in (0, 1):
            keep = eng.stretch_split(sp, rec["labels"], rec[f"rint{sp}"], rec[f"u_zz{sp}"], rec[f"u_acc{sp}"])
            knife = np.abs(rec[f"lnpdiff{sp}"] - np.log(rec[f"u_acc{sp}"])) < 1e-12
            assert not knife.any(), "a decision on the knife edge: pick another seed for this fixture"
            assert np.array_equal(keep, rec[f"keep{sp}"]), f"iteration {it} split {sp}: accept mask"
            x, L, P, _ = eng.download()
            assert np.array_equal(x, rec[f"x_after{sp}"]), f"iteration {it} split {sp}: positions"
        tol.check_logl(L, rec["L_stretch"], what="log-like after the MH step")
        assert np.array_equal(P, rec["P_stretch"])
    assert nst >= 4
    eng.close()


def test_philox_stepping_with_periodic_parameters_stays_on_the_circle():
    """hens_step (Philox) with periodic parameters runs through the generic-width kernel: every accepted proposal is
    wrapped (stretch.py:149-154, gaussian.py:110-115), and a von-Mises-like target on the circle is sampled without
    the seam at 0 / period showing (distance the short way round, stretch.py:136-141)."""
    from eryn_amd.engine import HipEnsemble
    T, W, D = 2, 2048, 4
    per = np.array([2 * np.pi, 0.0, 0.0, 0.0])
    # likelihood centred ON the seam of the periodic parameter: without periodic distances / wrapping the walkers on the two
    # sides of 0 = 2 pi would be 2 pi apart
    mu = np.array([0.05, 0.0, 0.0, 0.0])
    like = GaussianLikelihood(mu, np.diag([4.0, 1.0, 1.0, 1.0]))
    eng = HipEnsemble(T, W, D, like, np.array([-0.5, -50, -50, -50.0]), np.array([2 * np.pi + 0.5, 50, 50, 50.0]), seed=7)
    eng.set_periodic(per)
    x0 = np.random.RandomState(2).randn(T, W, D) * 0.3
    x0[..., 0] = np.random.RandomState(3).uniform(0.0, 2 * np.pi, size=(T, W))
    eng.upload(x0, betas=np.array([1.0, 0.5]))
    eng.eval_state()
    eng.set_mh_proposal("iso", 0.3, 0.3)
    eng.step(300)
    x, L, P, _ = eng.download()
    assert np.all((x[..., 0] >= 0.0) & (x[..., 0] < 2 * np.pi))
    assert np.isfinite(L).all() and np.isfinite(P).all()
    c = eng.counters()
    assert c["accepted"].sum() > 0 and eng.mh_counters()["accepted"].sum() > 0
    eng.set_periodic(None)                                     # back to the compile-time-width kernels
    eng.step(3)
    eng.close()


@pytest.mark.parametrize("name", ["m1_gauss_iso", "m3_gauss_full", "m8_mix_narrowbox", "p2_mix_periodic"])
def test_mh_step_teacher_forced_against_oracle(name, golden_dir):
    """hens_mh_step through the C ABI, one proposal at a time, against the oracle's intermediates."""
    from eryn_amd.engine import HipEnsemble
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    T, W, D, box = int(fx["T"]), int(fx["W"]), int(fx["D"]), float(fx["box"])
    o = build_mh_oracle(fx)
    eng = HipEnsemble(T, W, D, GaussianLikelihood(fx["mu"], fx["invcov"]), -box, box, tempered=o.tempered)
    if "period" in fx.files:
        eng.set_periodic(fx["period"])                         # gaussian.py:110-115: q = wrap(x + step)
    for it in range(int(fx["nsteps"])):
        prev = (o.x.copy(), o.L.copy(), o.P.copy(), None if o.betas is None else o.betas.copy())
        o.iteration()
        rec = o.trace[-1]
        if "mh_step" not in rec:
            continue
        eng.upload(*prev)
        keep = eng.mh_step(rec["mh_step"], rec["mh_u_acc"])
        knife = np.abs(rec["mh_lnpdiff"] - np.log(rec["mh_u_acc"])) < 1e-12
        assert np.array_equal(keep | knife, rec["mh_keep"] | knife)
        x, L, P, _ = eng.download()
        ok = ~knife
        assert np.array_equal(x[ok], np.where(rec["mh_keep"][..., None], rec["mh_q"], prev[0])[ok])
        tol.check_logl(L[ok], rec["L_stretch"][ok], what="log-like after the periodic step")
        assert np.array_equal(P[ok], rec["P_stretch"][ok])


def test_mh_shapes_sizes_and_errors():
    from eryn_amd.engine import HipEnsemble
    like = GaussianLikelihood(np.zeros(5), np.eye(5))
    eng = HipEnsemble(2, 70, 5, like, -3.0, 3.0)               # odd W, generic row width
    x0 = np.random.RandomState(0).uniform(-1, 1, size=(2, 70, 5))
    eng.upload(x0, betas=np.array([1.0, 0.5]))
    eng.eval_state()
    step = np.zeros((2, 70, 5))
    keep = eng.mh_step(step, np.full((2, 70), 0.5))            # q = x: lnpdiff = 0 > log 0.5 -> all accepted
    assert keep.all()
    x, _, _, _ = eng.download()
    assert np.array_equal(x, x0)
    step[...] = 100.0                                          # leaves the box: never accepted (prior.py:80-88)
    assert not eng.mh_step(step, np.full((2, 70), 0.5)).any()
    with pytest.raises(ValueError):
        eng.mh_step(np.full((2, 70, 5), np.nan), np.full((2, 70), 0.5))
    with pytest.raises(ValueError):
        GaussianMove({"model_0": 0.1}, factor=0.5)
    with pytest.raises(ValueError):
        GaussianMove({"model_0": np.eye(3)}, mode="random")


@pytest.mark.parametrize("kind", ["iso", "diag", "full"])
def test_philox_gaussian_move_targets_the_gaussian(kind):
    """Device-side Box-Muller draws: a pure MH sampler and a stretch+MH mix must sample the analytic target."""
    T, W, D = 2, 4096, 4
    rs = np.random.RandomState(0)
    A = rs.randn(D, D)
    mu = 0.1 * rs.randn(D)
    cov = A @ A.T / D + np.eye(D)
    prop = {"iso": 0.3, "diag": np.array([0.2, 0.3, 0.4, 0.3]), "full": 0.3 * cov}[kind]
    priors = {i: uniform_dist(-50.0, 50.0) for i in range(D)}
    for moves in ([GaussianMove({"model_0": prop})],
                  [(StretchMove(), 0.5), (GaussianMove({"model_0": prop}), 0.5)]):
        s = EnsembleSampler(W, D, GaussianLikelihood(mu, np.linalg.inv(cov)), priors, moves=moves,
                            tempering_kwargs=dict(ntemps=T), rng="philox", seed=5)
        x0 = np.random.RandomState(1).randn(T, W, D)
        s.run_mcmc(x0, 10, burn=400, thin_by=20)
        chain = s.get_chain()["model_0"][:, 0, :, 0, :].reshape(-1, D)
        assert np.abs(chain.mean(0) - mu).max() < 0.06
        assert np.linalg.norm(np.cov(chain.T) - cov) / np.linalg.norm(cov) < 0.06
        for m in s.moves:
            assert m.num_proposals > 0
            frac = (m.accepted / m.num_proposals)[0].mean()
            assert 0.1 < frac < 0.95
        assert sum(m.num_proposals for m in s.moves) == 400 + 200


def test_philox_normals_are_standard():
    """Moments of the device Box-Muller stream (one proposal from a point mass = the step itself)."""
    from eryn_amd.engine import HipEnsemble
    D, W = 8, 8192
    like = GaussianLikelihood(np.zeros(D), 1e-12 * np.eye(D))  # flat: every proposal is accepted
    eng = HipEnsemble(1, W, D, like, -1e6, 1e6, tempered=False, seed=3)
    eng.upload(np.zeros((1, W, D)))
    eng.eval_state()
    eng.set_mh_proposal("iso", 1.0, 1.0)
    eng.step(1)
    z, _, _, _ = eng.download()
    z = z.reshape(-1)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    assert abs(np.mean(z ** 3)) < 0.05 and abs(np.mean(z ** 4) - 3.0) < 0.1
    assert abs(np.corrcoef(z[0::2], z[1::2])[0, 1]) < 0.02     # the two outputs of one Box-Muller pair
