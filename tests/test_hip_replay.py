"""The code path the benchmark times, held to the oracle (-m gpu).

``hens_step`` (Philox mode: plan kernel, inline-permutation cascade, ladder adaptation folded into the next
launch) is replayed through ``oracle/eryn_oracle.py`` with the very draws it consumed (``hens_debug_draws`` ->
``tests/replay_utils.py``): positions, log-prior, accept and swap counters exact; log-likelihood rtol 1e-12; betas
rtol 1e-13.  Every case runs at least one call of >= 3 iterations so that the folded adaptation (a pending
adaptation riding in the next iteration's first launch) is on the replayed path."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import eryn_oracle as orc
from tests import parity_utils as pu
from tests import replay_utils as ru

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _engine(T, W, D, like, box, seed, **kw):
    from eryn_amd.engine import HipEnsemble
    return HipEnsemble(T, W, D, like, -box, box, seed=seed, **kw)


def _run_case(T, W, D, like_kind="dense", box=50.0, seed=77, calls=(1, 3), x_scale=1.0, mh=None, period=None, nsplits=2, **kw):
    from eryn_amd.likelihood import GaussianLikelihood, RosenbrockLikelihood
    mu, invcov = pu.gaussian_problem(D, dense=(like_kind == "dense"))
    if like_kind == "dense":
        like, fn = GaussianLikelihood(mu, invcov), (lambda x: orc.gaussian_log_like(x, mu, invcov))
    elif like_kind == "diag":
        iv = np.diag(invcov).copy()
        like, fn = GaussianLikelihood(mu, iv), (lambda x: orc.gaussian_diag_log_like(x, mu, iv))
    else:
        like, fn = RosenbrockLikelihood(D), (lambda x: orc.rosenbrock_log_like(x))
    eng = _engine(T, W, D, like, box, seed, **kw)
    lo, hi = np.full(D, -box), np.full(D, box)
    # inside the prior support: the reference refuses a start with an infinite log-prior (ensemble.py:930-946)
    x0 = np.clip(x_scale * np.random.RandomState(3).randn(T, W, D), -0.95 * box, 0.95 * box)
    tempered = T > 1
    eng.upload(x0, betas=orc.make_ladder(D, ntemps=T) if tempered else None)
    eng.eval_state()
    if mh is not None:
        eng.set_mh_proposal(*mh)
    if period is not None:
        eng.set_periodic(period)
    if nsplits != 2:
        eng.set_nsplits(nsplits)
    x, L, P, betas = eng.download()
    st = ru.OracleState(x, L, P, betas, time=0)
    kinds = []
    done = 0
    for n in calls:
        it0 = eng.iteration()
        eng.step(n)
        eng.synchronize()
        kinds += ru.replay(eng, st, it0, n, fn, lo, hi, mh=mh is not None, period=period, nsplits=nsplits,
                           adaptive=kw.get("adaptive", True), stop_adaptation=kw.get("stop_adaptation", -1))
        x, L, P, betas = eng.download()
        ru.assert_state_equal(st, x, L, P, betas, counters=eng.counters(),
                              mh_counters=eng.mh_counters() if mh is not None else None,
                              what=f"({T},{W},{D}) {like_kind} after {done + n} iterations")
        done += n
        assert st.min_margin > 1e-12, "a decision sat on the knife edge; pick another seed for this case"
    acc = st.accepted.sum() + st.mh_accepted.sum()
    assert acc > 0, "nothing was ever accepted: the case does not exercise the update"
    if tempered:
        assert st.swaps_total.sum() > 0
    eng.close()
    return kinds


def test_replay_config2_full_size():
    """BASELINE config 2 at full size: one single-iteration call, then a 3-iteration call (folded adaptation)."""
    _run_case(16, 4096, 32)


def test_replay_ndim64():
    _run_case(8, 1024, 64, calls=(4,))


def test_replay_long_ladder_64_rungs():
    """config 3's ladder length at reduced W: the 63-pair cascade and the multi-word swap bitmask."""
    _run_case(64, 1024, 64, calls=(2, 3))


def test_replay_rosenbrock_ndim128_move_mix():
    """BASELINE config 5 in small: Rosenbrock, StretchMove + GaussianMove mixed by weight (ensemble.py:971)."""
    kinds = _run_case(4, 512, 128, like_kind="rosen", box=6.0, calls=(3, 5), x_scale=0.5, mh=("iso", 0.01, 0.5))
    assert "mh" in kinds and "stretch" in kinds


def test_replay_diag_move_mix_axis_aligned():
    kinds = _run_case(3, 256, 16, like_kind="diag", box=8.0, calls=(6,), mh=("diag", np.full(16, 0.3), 0.4))
    assert "mh" in kinds and "stretch" in kinds


@pytest.mark.parametrize("T,W,D,like,weight", [(4, 512, 32, "dense", 0.5), (8, 256, 16, "dense", 1.0), (3, 200, 6, "diag", 0.6)])
def test_replay_full_covariance_move_mix(T, W, D, like, weight):
    """GaussianMove with a FULL covariance in production (gaussian.py:265-268: multivariate normal steps): k_mh_draw's
    Box-Muller normals times the Cholesky factor (staged through LDS in the production launch), the step buffer and the MH
    launch - held to the oracle like the isotropic / axis-aligned forms (a compile-time row width in record mode, a pure MH
    chain, and a padded generic width)."""
    rs = np.random.RandomState(21)
    a = rs.randn(D, D)
    chol = np.linalg.cholesky(0.02 * (a @ a.T / D + np.eye(D)))
    kinds = _run_case(T, W, D, like_kind=like, box=20.0, calls=(3, 4), x_scale=0.7, mh=("full", chol, weight))
    assert "mh" in kinds and (weight >= 1.0 or "stretch" in kinds)


@pytest.mark.parametrize("T,W,D", [(3, 257, 8), (3, 33, 4), (5, 100, 5)])
def test_replay_odd_sizes(T, W, D):
    """odd W (halves of ceil / floor size), W not a multiple of the tile, generic row widths"""
    _run_case(T, W, D, calls=(1, 4))


@pytest.mark.parametrize("T,W,D", [(16, 40, 8), (8, 48, 16), (32, 20, 8), (64, 16, 8), (2, 64, 8)])
def test_replay_small_two_launch_shapes(T, W, D):
    """block-balanced shapes with fewer walkers per half than a tile, few column blocks, cb = 2 ... 64"""
    _run_case(T, W, D, calls=(2, 3), x_scale=0.5)


def test_replay_untempered():
    _run_case(1, 64, 5, like_kind="diag", calls=(4,))                  # generic row width: copying launches
    _run_case(1, 256, 16, calls=(2, 3))                                # compile-time row width: rows updated in place
    kinds = _run_case(1, 512, 32, calls=(3, 4), mh=("iso", 0.3, 0.5))   # ... with the MH move in the mix
    assert "mh" in kinds and "stretch" in kinds


def test_replay_narrow_box():
    """many proposals outside the prior support: -inf prior, fill value, P inf -> 0 rule (move.py:526)"""
    _run_case(4, 256, 16, box=2.0, calls=(2, 3), x_scale=0.8)


def test_replay_adaptation_stops():
    _run_case(4, 128, 8, calls=(5,), stop_adaptation=2)
    _run_case(4, 128, 8, calls=(4,), adaptive=False)


@pytest.mark.parametrize("T,W,D,like", [(8, 4096, 32, "dense"), (16, 2048, 16, "dense"), (4, 512, 32, "diag"), (32, 256, 32, "rosen")])
def test_replay_one_launch_shapes(T, W, D, like):
    """shapes that step in ONE launch per iteration (k_iter: replayed complements, versioned rows, three count buffers in
    rotation - four consecutive iterations in a call cover a whole turn of them)"""
    _run_case(T, W, D, like_kind=like, box=6.0 if like == "rosen" else 50.0, calls=(1, 4, 2), x_scale=0.5 if like == "rosen" else 1.0)


def test_replay_one_launch_variants():
    """k_iter with the adaptation stopping / switched off (the launch still carries the swap counters), on the 64-rung
    ladder (two-word swap masks, one column pair per workgroup) and with the Metropolis-Hastings move in the mix (its
    cascade's counts are adapted by a kernel of their own before the next single launch)"""
    _run_case(4, 128, 16, calls=(5,), stop_adaptation=2)
    _run_case(8, 64, 32, calls=(4,), adaptive=False)
    _run_case(64, 64, 32, calls=(2, 3), x_scale=0.5)
    kinds = _run_case(8, 256, 32, calls=(3, 6), mh=("iso", 0.3, 0.5))
    assert "mh" in kinds and "stretch" in kinds


@pytest.mark.parametrize("T,W,D,like,mh", [(5, 100, 5, "dense", None), (8, 2048, 11, "dense", None),
                                           (4, 256, 24, "diag", ("diag", None, 0.4)), (2, 512, 12, "dense", ("iso", 0.3, 0.5)),
                                           (4, 144, 70, "dense", None)])
def test_replay_padded_rows(T, W, D, like, mh):
    """Row widths without a compile-time-width kernel are padded to the next one (engine.padded_width: zeros under a
    (-inf, +inf) prior interval and zero rows of the precision matrix; hens_config::ndim_active keeps the Hastings
    factor (D - 1) log zz on the real dimension): hens_step through the oracle, which knows nothing of the pads."""
    from eryn_amd.engine import FAST_WIDTHS, padded_width
    if mh is not None and mh[1] is None:
        mh = (mh[0], np.full(D, 0.2), mh[2])
    kinds = _run_case(T, W, D, like_kind=like, calls=(1, 4), mh=mh)
    assert "stretch" in kinds

    class K:
        kind = 0
    assert padded_width(D, K) in FAST_WIDTHS and padded_width(D, K) >= D


@pytest.mark.parametrize("T,W,D,mh", [(10, 4096, 32, None), (12, 512, 64, None), (20, 2048, 32, None), (24, 256, 8, None),
                                      (3, 4096, 32, None), (5, 2048, 16, None), (6, 256, 32, ("iso", 0.3, 0.5)),
                                      (10, 256, 16, None), (48, 64, 32, None), (7, 256, 128, None), (33, 130, 16, None)])
def test_replay_ladders_that_do_not_divide_128(T, W, D, mh):
    """Block-balanced labels with cb = the largest power of two with cb T <= 128 (ntemps = 10, 20, ... are what people
    use): short tiles - cb T / 2 < 64 moving walkers per workgroup - through the two-launch iteration (k_split1_pt) and,
    for the small shapes, the one-launch iteration (k_iter); 33+ rungs take two-word swap masks."""
    kinds = _run_case(T, W, D, calls=(1, 4), mh=mh, x_scale=0.7)
    assert "stretch" in kinds


@pytest.mark.parametrize("T,W,D,mh", [(3, 33, 4, None), (5, 100, 5, None), (4, 128, 12, ("iso", 0.3, 0.5))])
def test_replay_generic_width_kernel_unpadded(T, W, D, mh):
    """the generic-width kernel itself (pad_rows=False: what Rosenbrock / host-likelihood contexts and widths above 128
    run), three copying launches per iteration"""
    _run_case(T, W, D, calls=(1, 4), mh=mh, pad_rows=False)


@pytest.mark.parametrize("T,W,D,like,mh", [(16, 4096, 32, "dense", None), (8, 256, 16, "dense", ("iso", 0.4, 0.5)),
                                           (8, 512, 32, "dense", None), (10, 256, 11, "dense", None),
                                           (4, 512, 64, "dense", None), (4, 256, 128, "dense", ("iso", 0.05, 0.5)),
                                           (3, 130, 6, "diag", ("diag", None, 0.4))])
def test_replay_periodic_parameters(T, W, D, like, mh):
    """hens_step with periodic parameters through the oracle: the compile-time-width kernels of the two-launch iteration
    (distances the short way round, wrapped proposals: stretch.py:136-154; MH proposals wrapped: gaussian.py:110-115) and
    the generic-width kernel.  A third of the parameters is periodic with periods small enough that the walkers spread
    over more than half of them."""
    period = np.zeros(D)
    period[::3] = np.linspace(1.5, 4.0, len(period[::3]))
    if mh is not None and mh[1] is None:
        mh = (mh[0], np.full(D, 0.2), mh[2])
    kinds = _run_case(T, W, D, like_kind=like, box=6.0 if like == "rosen" else 50.0, calls=(1, 3),
                      x_scale=0.5 if like == "rosen" else 1.0, mh=mh, period=period)
    assert "stretch" in kinds


@pytest.mark.parametrize("nranks", [2, 4])
def test_replay_local_pipeline(tmp_path, nranks):
    """N ladder shards stepping through the pipeline (one-sided puts, per-block hand-off flags) against the oracle"""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", HENS_PIPE_TIMEOUT_S="10")
    r = subprocess.run([sys.executable, os.path.join(HERE, "pipeline_worker.py"), "replay", str(nranks), "8", "256", "32", "5"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("T,W,D,nsplits,mh", [(3, 67, 8, 3, None), (4, 1000, 32, 5, None), (1, 130, 16, 4, None),
                                              (8, 512, 64, 3, ("iso", 0.3, 0.4)), (2, 257, 5, 8, None), (16, 4096, 32, 3, None)])
def test_replay_red_blue_move_of_more_than_two_sets(T, W, D, nsplits, mh):
    """RedBlueMove(nsplits = n > 2) with device draws (red_blue.py:41-47,119-124,148-197; VERDICT r3 "missing" 3): hens_step
    labels every rung with a keyed permutation mod n (set k: ceil((W - k) / n) walkers, ascending), moves set after set, each
    against the other sets concatenated in set order, and the oracle - whose n-set form is pinned to the reference by fixture
    f8_nsplits3 - replays the very iterations with the exported draws: odd walker counts, padded and generic row widths, an
    untempered ensemble, the MH move in the mix, config 2's shape."""
    kinds = _run_case(T, W, D, nsplits=nsplits, mh=mh, calls=(1, 3) if mh is None else (2, 6))
    if mh is not None:
        assert "mh" in kinds and "stretch" in kinds
