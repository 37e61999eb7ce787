"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle.

Teacher-forced per step (SURVEY 7 hard part 4): every iteration starts from the oracle's
state and uses the oracle's (= the reference's) random draws, so a single knife-edge flip
cannot snowball.  Tolerances:
  accept / swap masks, swap counts, indices ... exact (knife-edge allowance 1e-12 relative,
                                               counted and expected to be 0)
  walker positions, log-prior ............... exact (proposal arithmetic is compiled without FMA)
  log-likelihood ............................ rtol 1e-13 (summation order of the quadratic form; observed: tolerance_log)
  betas after adaptation .................... rtol 1e-13 (device exp vs libm exp)
"""
import os

import numpy as np
import pytest

from tests import parity_utils as pu
from tests import tolerance_log as tol
from oracle import eryn_oracle as orc

pytestmark = pytest.mark.gpu


def _fixture_oracle(fx):
    T, W, D = int(fx["T"]), int(fx["W"]), int(fx["D"])
    R = np.random.RandomState(int(fx["seed_construct"]))
    G = np.random.RandomState(int(fx["seed_run"]))
    box = float(fx["box"])
    mu, invcov = fx["mu"], fx["invcov"]
    kw = {}
    if "betas0" in fx.files:
        kw.update(betas=fx["betas0"], adaptive=bool(fx["adaptive"]), permute=bool(fx["permute"]))
    if "nsplits" in fx.files:
        kw["nsplits"] = int(fx["nsplits"])
    o = orc.OracleSampler(fx["x0"], lambda x: orc.gaussian_log_like(x, mu, invcov), np.full(D, -box),
                          np.full(D, box), R, G, a=float(fx["a"]), record=True, **kw)
    return o, mu, invcov


@pytest.mark.parametrize("name", ["f1_plumbing", "f2_pt", "f3_oddW", "f4_narrowbox", "f5_noadapt",
                                  "f5_nopermute", "f6_medium", "f7_tmaxinf", "f8_nsplits3"])
def test_golden_fixture_teacher_forced(name, golden_dir):
    """Replay the reference's own recorded draws (committed fixtures) through the HIP path."""
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    o, mu, invcov = _fixture_oracle(fx)
    dense = bool(fx["dense"])
    eng = pu.make_engine(o, mu, invcov, dense=True)
    n = int(fx["nsteps"])
    tolerated = 0
    for it in range(n):
        prev = (o.x.copy(), o.L.copy(), o.P.copy(), None if o.betas is None else o.betas.copy(), o.time)
        o.iteration()
        rec = o.trace[-1]
        # the oracle trace is pinned to the fixture on CPU; cross-check the masks against the file here too
        for sp in range(o.nsplits):
            assert np.array_equal(rec[f"keep{sp}"], fx[f"it{it}_keep{sp}"])
        tolerated += pu.check_iteration(eng, o, rec, prev, teacher_forced=True)
        o.trace.clear()
    assert tolerated == 0
    eng.close()


@pytest.mark.parametrize("T,W,D,box,n", [
    (16, 4096, 32, 50.0, 3),      # BASELINE config 2, full size
    (4, 1024, 64, 50.0, 2),       # D = 64 kernel (config 3 row width)
    (8, 16384, 64, 50.0, 1),      # BASELINE config 3, one GPU's shard at full size (8 of the 64 rungs)
    (4, 512, 128, 50.0, 2),       # config 4's row width (generic kernel)
    (3, 130, 7, 2.0, 6),          # odd D (scalar rows), uneven tiles, narrow box
    (2, 64, 16, 50.0, 4),
    (5, 256, 8, 1.5, 4),
    (1, 128, 32, 50.0, 4),        # no tempering
])
def test_seeded_teacher_forced(T, W, D, box, n):
    o, mu, invcov = pu.make_oracle(T, W, D, box=box, x0=np.random.RandomState(1).uniform(-0.9 * min(box, 3.0), 0.9 * min(box, 3.0), size=(T, W, D)))
    eng = pu.make_engine(o, mu, invcov)
    stats = {}
    tolerated = pu.run_parity(o, eng, n, teacher_forced=True, stats=stats)
    assert tolerated == 0
    assert stats.get("max_rel_L", 0.0) <= tol.RTOL_L
    eng.close()


def test_free_running_short():
    """No re-upload between iterations: device state carries over (loc indirection, double buffers)."""
    o, mu, invcov = pu.make_oracle(4, 256, 8, box=50.0)
    eng = pu.make_engine(o, mu, invcov)
    tolerated = pu.run_parity(o, eng, 6, teacher_forced=False)
    assert tolerated == 0
    eng.close()


def test_eval_state_matches_oracle():
    T, W, D = 3, 200, 32
    o, mu, invcov = pu.make_oracle(T, W, D, box=1.0, x0=np.random.RandomState(3).uniform(-1.3, 1.3, size=(T, W, D)))
    eng = pu.make_engine(o, mu, invcov)
    eng.upload(o.x, betas=o.betas)
    eng.eval_state()
    _, L, P, _ = eng.download()
    assert np.array_equal(P, o.P)
    assert np.array_equal(L == -1e300, o.L == -1e300)
    tol.check_logl(L, o.L, what="hens_eval_state")
    eng.close()


@pytest.mark.parametrize("D", [16, 32, 64, 128, 20])     # scalar-operand form / matrix pipe at three widths / rows padded to 32
def test_eval_state_nonsymmetric_precision_and_walkers_outside_the_box(D):
    """The device packs A_ik + A_ki (pair layout) or M_IJ = A_IJ + A_JI^T (matrix-pipe operands, hens_set_gaussian): a precision matrix
    that is not symmetric gives the oracle's quadratic form (x - mu)^T A (x - mu) all the same.  Walkers outside the prior box - some of
    them with huge coordinates - ride along in the tile (a walker is a column of the block products) and must not disturb their
    neighbours: they get the fill value (ensemble.py:1486-1513), everybody else the oracle's value."""
    T, W = 2, 320
    R = np.random.RandomState(7)
    mu, invcov = pu.gaussian_problem(D, True)
    invcov = invcov + 0.05 * np.triu(R.randn(D, D), 1) * np.abs(invcov).mean()       # upper triangle only: not symmetric
    x0 = R.uniform(-0.9, 0.9, size=(T, W, D))
    out = R.rand(T, W) < 0.2
    x0[out, R.randint(0, D, size=int(out.sum()))] = R.choice([1.5, -3.0, 1e300, -1e308], size=int(out.sum()))
    o = orc.OracleSampler(x0, lambda x: orc.gaussian_log_like(x, mu, invcov), np.full(D, -1.0), np.full(D, 1.0),
                          np.random.RandomState(1), np.random.RandomState(2), betas=orc.make_ladder(D, ntemps=T), record=False)
    eng = pu.make_engine(o, mu, invcov)
    eng.upload(o.x, betas=o.betas)
    eng.eval_state()
    _, L, P, _ = eng.download()
    assert np.array_equal(P, o.P) and np.array_equal(np.isinf(P), out)
    assert np.array_equal(L == -1e300, o.L == -1e300) and np.array_equal(L == -1e300, out)
    tol.check_logl(L, o.L, what=f"hens_eval_state, non-symmetric precision, D = {D}")
    eng.close()


@pytest.mark.parametrize("T,W,D", [(2, 128, 5), (2, 256, 32), (3, 512, 128)])     # generic / fast / D = 128 fast kernels
def test_diag_likelihood(T, W, D):
    o, mu, invcov = pu.make_oracle(T, W, D, box=5.0, dense=False)
    eng = pu.make_engine(o, mu, invcov, dense=False)
    assert pu.run_parity(o, eng, 5, teacher_forced=True) == 0
    eng.close()


def test_error_mapping():
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import GaussianLikelihood
    like = GaussianLikelihood(np.zeros(8), np.eye(8))
    with pytest.raises(RuntimeError, match="fewer walkers"):
        HipEnsemble(1, 10, 8, like, -1, 1)                      # red_blue.py:108-114
    eng = HipEnsemble(1, 10, 8, like, -1, 1, live_dangerously=True)
    with pytest.raises(RuntimeError):
        eng.step(1)                                             # no state yet
    x = np.zeros((1, 10, 8))
    x[0, 3, 2] = np.nan
    eng.upload(x)
    with pytest.raises(ValueError):
        eng.eval_state()                                        # ensemble.py:1258-1262
    eng.close()


@pytest.mark.parametrize("W", [2, 7, 64, 100, 4096, 5000])
def test_philox_permutations_are_uniform_bijections(W):
    """The production RNG pairs walkers through a keyed Feistel permutation instead of sorting random
    keys: it must be a bijection for every key and look uniform (position frequencies, fixed points,
    displacement) over many keys."""
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import GaussianLikelihood
    eng = HipEnsemble(2, W, 1, GaussianLikelihood(np.zeros(1), np.eye(1)), -1, 1, seed=99, live_dangerously=True)
    n = 400 if W <= 100 else 60
    perms = np.stack([eng.debug_permutation(k % 2, k % 2, k) for k in range(n)])
    assert np.array_equal(np.sort(perms, axis=1), np.tile(np.arange(W), (n, 1))), "not a bijection"
    if W >= 7:
        fixed = (perms == np.arange(W)).sum(axis=1)                 # ~Poisson(1) for uniform permutations
        assert 0.5 < fixed.mean() < 1.6
        # mean |perm(c) - c| of a uniform permutation is (W^2 - 1) / (3 W)
        disp = np.abs(perms - np.arange(W)).mean()
        assert abs(disp / ((W * W - 1) / (3.0 * W)) - 1.0) < 0.05
        # no position is favoured: chi-square of where element 0 and element W-1 land (coarse bins)
        for el in (0, W - 1):
            nb = W if W <= 100 else 8                               # landing-position histogram
            edges = np.linspace(0, W, nb + 1)
            obs = np.histogram(perms[:, el], bins=edges)[0]
            exp = n * np.diff(np.ceil(edges)) / W
            assert np.all(np.abs(obs - exp) < 5.0 * np.sqrt(exp) + 1.0), (obs, exp)
        # successive keys are unrelated
        assert np.mean(perms[0] == perms[1]) < 0.5
    eng.close()


@pytest.mark.parametrize("T,W,D", [(3, 128, 8), (2, 100, 5), (2, 256, 32), (3, 512, 128)])
def test_rosenbrock_likelihood(T, W, D):
    """The config-5 stress likelihood (not in the reference; defined in oracle.rosenbrock_log_like)
    through the same fused kernel, teacher-forced: low acceptance, generic and fast row widths."""
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import RosenbrockLikelihood
    box = 3.0
    R, G = np.random.RandomState(5), np.random.RandomState(6)
    x0 = np.random.RandomState(1).uniform(-1.5, 1.5, size=(T, W, D))
    o = orc.OracleSampler(x0, lambda x: orc.rosenbrock_log_like(x, 1.0, 100.0), np.full(D, -box), np.full(D, box), R, G,
                          betas=orc.make_ladder(D, ntemps=T), record=True)
    eng = HipEnsemble(T, W, D, RosenbrockLikelihood(D, 1.0, 100.0), -box, box)
    stats = {}
    assert pu.run_parity(o, eng, 5, teacher_forced=True, stats=stats) == 0
    assert 0.0 < o.accepted.mean() / o.num_proposals < 0.6
    eng.close()


def test_stretch_scale_can_change_between_proposals():
    """StretchMove.a is a plain attribute in the reference (stretch.py:37, read per proposal at :129-132) and its tuning hook
    mutates it (utils/updates.py:130-175): hens_set_stretch_scale.  Teacher-forced against the oracle with a = 2, then 3, then
    1.5; and hens_step's Philox iterations replayed through the oracle with the scale they ran at."""
    from tests import replay_utils as ru
    T, W, D = 4, 512, 32
    o, mu, invcov = pu.make_oracle(T, W, D, box=50.0)
    eng = pu.make_engine(o, mu, invcov)
    for a in (2.0, 3.0, 1.5):
        o.a = a
        eng.set_stretch_scale(a)
        assert pu.run_parity(o, eng, 2, teacher_forced=True) == 0
    eng.upload(o.x, o.L, o.P, o.betas)
    st = ru.OracleState(o.x, o.L, o.P, o.betas, time=o.time)
    eng.set_adapt_time(o.time)
    for a in (2.5, 1.7):
        eng.set_stretch_scale(a)
        it0 = eng.iteration()
        eng.step(2)
        ru.replay(eng, st, it0, 2, lambda q: orc.gaussian_log_like(q, mu, invcov), o.lo, o.hi, a=a)
    x, L, P, betas = eng.download()
    ru.assert_state_equal(st, x, L, P, betas, what="hens_step after hens_set_stretch_scale")
    with pytest.raises(ValueError):
        eng.set_stretch_scale(1.0)
    eng.close()
