"""Repeatability soak of the production path (-m gpu): the same seed twice must give the same chain bit for bit.

The oracle replays (test_hip_replay.py) cover a handful of iterations per case; an ordering bug that strikes once in 10^4 - 10^5
launch boundaries is invisible to them.  Round 3 found one that way (`tools/soak_flaky.py`: the draw plan of batch b + 1 on a side
stream overlapped batch b's readers about once in 10^5 batch boundaries at 32 x 1024 x 16 - one chain in five of 10^5 iterations
parted ways with its own repeats).  Round 4 removed the side stream altogether (every plan runs on the context's one stream, in
front of the batch that reads it); these tests are the soak that would have caught it, inside the suite the driver runs: every
stepping path of `hens_step` - the two launches of config 2 in column order, the single launch `k_iter`, planned draws of a generic
width stepped in calls that cross the batches and the 1024-iteration key windows at odd offsets, and the shape that diverged - is
run twice from one seed in two fresh contexts and compared in every array and counter.  A data race between launches (in-place rows,
versioned rows, the scattered record stores of the column order, the swap-count rotation) would show as a difference as well.
"""
import numpy as np
import pytest

from oracle import eryn_oracle as orc
from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


def _chain(T, W, D, n_iters, call, seed=77, mh=None):
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import GaussianLikelihood
    mu, invcov = pu.gaussian_problem(D)
    eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=seed)
    eng.upload(np.random.RandomState(3).randn(T, W, D), betas=orc.make_ladder(D, ntemps=T) if T > 1 else None)
    eng.eval_state()
    if mh is not None:
        eng.set_mh_proposal(*mh)
    done = 0
    while done < n_iters:
        k = min(call, n_iters - done)
        eng.step(k)
        done += k
    eng.synchronize()
    x, L, P, betas = eng.download()
    c = eng.counters()
    out = dict(x=x, L=L, P=P, betas=betas, accepted=c["accepted"], swaps_total=c["swaps_total"], swaps_last=c["swaps_last"],
               num_proposals=np.array(c["num_proposals"]), adapt_time=np.array(c["adapt_time"]))
    if mh is not None:
        m = eng.mh_counters()
        out.update(accepted_mh=m["accepted"], num_mh=np.array(m["num_proposals"]))
    eng.close()
    return out


def _same(a, b, what):
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), f"{what}: `{k}` differs between two runs from the same seed"
    assert np.isfinite(a["x"]).all() and a["accepted"].sum() > 0


@pytest.mark.parametrize("shape", [
    # (T, W, D, iterations, iterations per call, MH mix, what)
    (16, 4096, 32, 20000, 20000, None, "config 2: two launches, column-ordered records"),
    (16, 4096, 32, 6000, 20, None, "config 2 in the driver's calls of 20"),
    (8, 4096, 32, 20000, 7777, None, "one launch per iteration (k_iter), planned draws"),
    (5, 100, 5, 20000, 7777, None, "generic width (padded rows), calls crossing batches and key windows"),
    (32, 1024, 16, 20000, 20000, None, "the shape whose side-stream plan diverged in round 3"),
    (32, 1024, 16, 20000, 333, ("iso", 0.05, 0.3), "the same with the MH move in the mix, short calls"),
    (8, 2048, 64, 6000, 1500, None, "D = 64: the likelihood on the matrix pipe"),
    (4, 1024, 128, 4000, 777, ("iso", 0.02, 0.4), "D = 128: matrix pipe, gathers in two halves, MH move in the mix"),
], ids=lambda s: f"{s[0]}x{s[1]}x{s[2]}-{s[4]}{'-mh' if s[5] else ''}")
def test_same_seed_same_chain(shape):
    T, W, D, n, call, mh, what = shape
    a = _chain(T, W, D, n, call, mh=mh)
    b = _chain(T, W, D, n, call, mh=mh)
    _same(a, b, what)


def test_split_into_calls_changes_nothing_over_a_long_run():
    """10^4 iterations in one call == the same in calls of 1 / 19 / 1000 iterations mixed (pack / unpack and the pending adaptation at
    every call boundary), on the single-launch shape and on config 2's two launches."""
    for T, W, D in ((8, 2048, 32), (16, 4096, 32)):
        a = _chain(T, W, D, 10000, 10000)
        from eryn_amd.engine import HipEnsemble
        from eryn_amd.likelihood import GaussianLikelihood
        mu, invcov = pu.gaussian_problem(D)
        eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=77)
        eng.upload(np.random.RandomState(3).randn(T, W, D), betas=orc.make_ladder(D, ntemps=T))
        eng.eval_state()
        done, k = 0, 0
        sizes = (1, 19, 1000, 2, 333)
        while done < 10000:
            n = min(sizes[k % len(sizes)], 10000 - done)
            eng.step(n)
            done += n
            k += 1
        x, L, P, betas = eng.download()
        c = eng.counters()
        eng.close()
        assert np.array_equal(a["x"], x) and np.array_equal(a["L"], L) and np.array_equal(a["P"], P)
        assert np.array_equal(a["betas"], betas) and np.array_equal(a["accepted"], c["accepted"])
        assert np.array_equal(a["swaps_total"], c["swaps_total"])
