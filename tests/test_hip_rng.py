"""Evidence for the production (Philox) randomness (-m gpu): the draws hens_step consumes, exported with
hens_debug_draws, against the distributions the reference draws from (red_blue.py:119-124, stretch.py:93-99,
tempering.py:526-535), and a config-2-scale run against the analytic tempered Gaussian."""
import numpy as np
import pytest

from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


def _engine(T, W, D, seed):
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import GaussianLikelihood
    mu, invcov = pu.gaussian_problem(D)
    return HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=seed)


def _chi2_z(counts, expected):
    """chi-square statistic as a z-score (large dof: chi2 ~ N(dof, 2 dof))."""
    chi2 = float(((counts - expected) ** 2 / expected).sum())
    dof = counts.size - 1
    return (chi2 - dof) / np.sqrt(2.0 * dof)


@pytest.mark.parametrize("T,W,iters", [(2, 64, 6000), (4, 100, 4000)])
def test_complement_and_matching_pairs_are_uniform(T, W, iters):
    """Every walker's complement is uniform over the OTHER walkers of its rung (uniform balanced split x uniform index),
    and the cascade matches slot a of rung i with slot b of rung i-1 uniformly; consecutive iterations' labels are
    independent.  ~1e4+ keys per case."""
    eng = _engine(T, W, 8, seed=123)
    pair = np.zeros((W, W))
    match = np.zeros((W, W))
    same_label = 0
    prev = None
    N0 = (W + 1) // 2
    for it in range(iters):
        d = eng.debug_draws(it)
        own, cw, slot = d["own"].astype(np.int64), d["cw"].astype(np.int64), d["pt_slot"].astype(np.int64)
        np.add.at(pair, (own.ravel(), cw.ravel()), 1)
        for i in range(1, T):
            np.add.at(match, (slot[i], slot[i - 1]), 1)
        lab = np.zeros((T, W), dtype=np.int8)
        lab[np.arange(T)[:, None], own[:, N0:]] = 1
        if prev is not None:
            same_label += int((lab == prev).sum())
        prev = lab
    eng.close()
    assert np.all(np.diag(pair) == 0)
    off = pair[~np.eye(W, dtype=bool)]
    n_picks = iters * T * W
    assert abs(_chi2_z(off, n_picks / (W * (W - 1)))) < 5.0, "complement pairs are not uniform"
    assert abs(_chi2_z(match.ravel(), iters * (T - 1) * W / W ** 2)) < 5.0, "PT matching is not uniform"
    # a walker keeps its label between two INDEPENDENT balanced labellings with probability P(0)^2 + P(1)^2 (the
    # binomial variance below is an upper bound: labels inside one labelling are negatively correlated)
    p_same = (N0 / W) ** 2 + ((W - N0) / W) ** 2
    n = (iters - 1) * T * W
    z = (same_label - n * p_same) / np.sqrt(n * p_same * (1 - p_same))
    assert abs(z) < 5.0, f"labels of consecutive iterations are correlated (z = {z:.1f})"


@pytest.mark.parametrize("T,W,iters", [(3, 16, 6000), (4, 24, 4000)])
def test_adjacent_pairs_matchings_are_independent(T, W, iters):
    """VERDICT r5 weak #1d.  The reference draws the matching of every pair of rungs independently (tempering.py:526-535: two fresh
    permutations per pair).  Production takes ONE keyed permutation per rung and per iteration - column c meets slot prp_t(c) of
    rung t - so the matching of pair (t, t-1) is M_t = prp_{t-1} o prp_t^{-1} and adjacent pairs SHARE a factor.  With
    independent uniform prp_t that is still a family of independent uniform matchings (given prp_t, M_t is uniform through
    prp_{t-1} and M_{t+1} through prp_{t+1}); this test looks for the dependence a flawed keying would leave:
    * the joint law of (partner above, partner below) of a middle-rung slot is uniform on W x W, for every slot (chi-square
      over W^3 cells),
    * the fixed-point counts of adjacent matchings (mean 1, variance 1 each) are uncorrelated,
    * and so are the lengths of the cycles through slot 0 - a statistic of the matching as a whole, not of one slot.
    (The SIGN of a matching is no such statistic here: see test_matchings_of_power_of_two_ensembles_are_even_permutations.)"""
    eng = _engine(T, W, 8, seed=4242)
    joint = np.zeros((T - 2, W, W, W))
    fp = np.zeros((iters, T - 1))
    cyc = np.zeros((iters, T - 1))

    def cycle_of_zero(p):
        j, n = int(p[0]), 1
        while j != 0:
            j, n = int(p[j]), n + 1
        return n

    for it in range(iters):
        slot = eng.debug_draws(it)["pt_slot"].astype(np.int64)          # [T][W]: the slot column c meets on rung t
        for t in range(1, T - 1):
            np.add.at(joint[t - 1], (slot[t], slot[t + 1], slot[t - 1]), 1)
        for t in range(1, T):
            m = np.empty(W, dtype=np.int64)
            m[slot[t]] = slot[t - 1]                                     # M_t: slot of rung t -> its partner on rung t - 1
            fp[it, t - 1] = (m == np.arange(W)).sum()
            cyc[it, t - 1] = cycle_of_zero(m)
    eng.close()
    for j in joint:
        assert abs(_chi2_z(j.ravel(), iters / W ** 2)) < 5.0, "(partner above, partner below) of a middle-rung slot is not uniform on W x W"
    for t in range(T - 2):
        for stat, what in ((fp, "fixed-point counts"), (cyc, "lengths of the cycle through slot 0")):
            a, b = stat[:, t], stat[:, t + 1]
            r = np.corrcoef(a, b)[0, 1]
            assert abs(r) * np.sqrt(iters) < 5.0, f"{what} of adjacent pairs' matchings are correlated (r = {r:.4f})"
    assert abs(fp.mean() - 1.0) < 5.0 / np.sqrt(fp.size)
    # the cycle through a given point of a uniform permutation of W points has a uniform length on 1 .. W (mean (W + 1) / 2)
    assert abs(cyc.mean() - (W + 1) / 2) < 5.0 * np.sqrt((W * W - 1) / 12.0 / cyc.size)


def test_matchings_of_power_of_two_ensembles_are_even_permutations():
    """A property found by the independence test above (round 6), stated here so that nobody has to find it again: a Feistel round
    x = (L, R) -> (L ^ f(R), R) is, for every R, 2^(lb - 1) disjoint transpositions - an EVEN permutation of 2^bits points for
    bits >= 2 - so the keyed column maps prp_t, and with them every matching prp_{t-1} o prp_t^-1, are even permutations whenever
    nwalkers is a power of two (no cycle walking): the matchings are drawn from the alternating group, the reference's from the
    whole symmetric group (tempering.py:526-535).  Harmless for the sampler - a swap sweep is valid for ANY matching chosen
    independently of the state, and every statistic of fewer than W - 1 slots (pair frequencies, joint partners, fixed points,
    cycle lengths: the tests above) is that of a uniform matching - and cheap to break (a keyed transposition on top) should a use
    ever care; recorded in DESIGN 7.  Ensembles that are not a power of two are cycle-walked and show both signs."""
    def sign(p):
        seen, s = np.zeros(len(p), dtype=bool), 1
        for i in range(len(p)):
            if not seen[i]:
                j, n = i, 0
                while not seen[j]:
                    seen[j] = True
                    j = p[j]
                    n += 1
                if n % 2 == 0:
                    s = -s
        return s

    for W, both in ((16, False), (24, True)):
        eng = _engine(3, W, 8, seed=99)
        signs = set()
        for it in range(200):
            slot = eng.debug_draws(it)["pt_slot"].astype(np.int64)
            m = np.empty(W, dtype=np.int64)
            m[slot[1]] = slot[0]
            signs.add(sign(m))
        eng.close()
        assert signs == ({1, -1} if both else {1}), (W, signs)


def test_large_ensemble_draws_binned():
    """W = 4096 (config 2's rung size): complement index differences and matching displacements in 64 bins each"""
    T, W, iters = 2, 4096, 300
    eng = _engine(T, W, 32, seed=77)
    diff = np.zeros(64)
    disp = np.zeros(64)
    u_all = []
    for it in range(iters):
        d = eng.debug_draws(it)
        own, cw, slot = d["own"].astype(np.int64), d["cw"].astype(np.int64), d["pt_slot"].astype(np.int64)
        diff += np.bincount((((cw - own) % W) * 64 // W).ravel(), minlength=64)
        disp += np.bincount((((slot[0] - slot[1]) % W) * 64 // W), minlength=64)
        u_all.append(np.concatenate([d["u_zz"].ravel()[::16], d["u_acc"].ravel()[::16], d["u_swap"].ravel()[::16]]))
    eng.close()
    # (cw - own) mod W is uniform over 1 .. W-1: bin 0 holds W/64 - 1 of the W - 1 values
    e = np.full(64, diff.sum() * (W / 64) / (W - 1))
    e[0] = diff.sum() * (W / 64 - 1) / (W - 1)
    assert abs(_chi2_z(diff, e)) < 5.0
    assert abs(_chi2_z(disp, disp.sum() / 64)) < 5.0
    u = np.concatenate(u_all)
    hist = np.bincount((u * 100).astype(int), minlength=100)[:100]
    assert abs(_chi2_z(hist, u.size / 100)) < 5.0 and 0.0 <= u.min() and u.max() < 1.0


def test_config2_scale_stationary_distribution():
    """16 x 4096 x 32 on a FIXED ladder (adaptive = False, so that every rung has one target for the whole run): every
    rung whose tempered Gaussian sits well inside the box has <logL> = -D / (2 beta) (x ~ N(mu, Sigma / beta)), the cold
    chain has the right mean and covariance, swap and accept rates are sane.  Errors from batch means over snapshots."""
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import GaussianLikelihood
    from eryn_amd.moves.tempering import make_ladder
    T, W, D = 16, 4096, 32
    mu, invcov = pu.gaussian_problem(D)
    cov = np.linalg.inv(invcov)
    betas0 = make_ladder(D, ntemps=T)
    eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024, adaptive=False)
    # start every rung from its own target (no long expansion phase for the hot rungs), then burn in
    rs = np.random.RandomState(1)
    Lc = np.linalg.cholesky(cov)
    x0 = mu + (rs.randn(T, W, D) @ Lc.T) / np.sqrt(betas0)[:, None, None]
    eng.upload(np.clip(x0, -49.0, 49.0), betas=betas0)
    eng.eval_state()
    eng.step(600)
    eng.reset_counters()
    nsnap, gap = 80, 25
    Lm = np.zeros((nsnap, T))
    xs = []
    for k in range(nsnap):
        eng.step(gap)
        x, L, P, betas = eng.download()
        Lm[k] = L.mean(axis=1)
        xs.append(x[0])
    c = eng.counters()
    eng.close()
    assert np.array_equal(betas, betas0)
    ok = np.sqrt(np.diag(cov).max() / betas) * 6 < 50.0          # rungs whose 6-sigma extent fits the +-50 box
    assert ok.sum() >= 8
    expect = -D / (2 * betas)
    # batch means: the snapshots are `gap` iterations apart; allow for residual correlation with a factor 2 on the error
    se = 2.0 * Lm.std(axis=0, ddof=1) / np.sqrt(nsnap)
    z = (Lm.mean(axis=0) - expect) / se
    assert np.all(np.abs(z[ok]) < 5.0), f"<logL> per rung off: z = {np.round(z, 1)} (ok = {ok})"
    rel = np.abs(Lm.mean(axis=0) - expect) / np.abs(expect)
    assert np.all(rel[ok] < 0.01), rel
    xc = np.concatenate(xs)
    assert np.abs(xc.mean(0) - mu).max() < 6 * np.sqrt(np.diag(cov).max() / (len(xc) / 20.0))
    relc = np.linalg.norm(np.cov(xc.T) - cov) / np.linalg.norm(cov)
    assert relc < 0.05, f"cold-chain covariance off by {relc:.3f}"
    frac = c["swaps_total"] / W / (nsnap * gap)
    assert np.all((frac > 0.05) & (frac < 0.7)), frac
    acc = c["accepted"].mean(axis=1) / c["num_proposals"]
    assert np.all((acc > 0.1) & (acc < 0.7)), acc
