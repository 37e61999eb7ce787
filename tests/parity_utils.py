"""Shared driver for the HIP-vs-oracle parity tests (also used by __graft_entry__.smoke)."""
import numpy as np

from oracle import eryn_oracle as orc


def gaussian_problem(D, dense=True):
    """SURVEY 8d synthetic Gaussian: mu = 0.1 randn, Sigma = A A^T / D + I (RandomState(0))."""
    rs = np.random.RandomState(0)
    A = rs.randn(D, D)
    mu = 0.1 * rs.randn(D)
    cov = A @ A.T / D + np.eye(D) if dense else np.eye(D)
    return mu, np.linalg.inv(cov)


def make_oracle(T, W, D, box=50.0, dense=True, seed_construct=123, seed_run=456, x0=None,
                tempered=None, record=True, **kw):
    mu, invcov = gaussian_problem(D, dense)
    R = np.random.RandomState(seed_construct)
    G = np.random.RandomState(seed_run)
    if x0 is None:
        x0 = np.random.RandomState(1).randn(T, W, D)
    if tempered is None:
        tempered = T > 1
    betas = orc.make_ladder(D, ntemps=T) if tempered else None
    if "betas" in kw:
        betas = kw.pop("betas")
    o = orc.OracleSampler(x0, lambda x: orc.gaussian_log_like(x, mu, invcov), np.full(D, -box), np.full(D, box),
                          R, G, betas=betas, record=record, **kw)
    return o, mu, invcov


def make_engine(o, mu, invcov, dense=True, **kw):
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import GaussianLikelihood
    like = GaussianLikelihood(mu, invcov if dense else np.diag(invcov).copy())
    return HipEnsemble(o.T, o.W, o.D, like, o.lo, o.hi, a=o.a, tempered=o.tempered, adaptive=o.adaptive,
                       adaptation_lag=o.lag, adaptation_time=o.nu, stop_adaptation=o.stop_adaptation, **kw)


from tests import tolerance_log as tol

KNIFE = 1e-12


def knife_edge(lnpdiff, logu):
    with np.errstate(invalid="ignore"):
        return np.abs(lnpdiff - logu) < KNIFE * np.maximum(1.0, np.abs(lnpdiff))


def check_iteration(eng, o, rec, prev, teacher_forced=True, rtol_l=None, stats=None):
    """Replay one recorded oracle iteration on the HIP engine and compare everything.

    prev = (x, L, P, betas, time) before the iteration (uploaded when teacher_forced).
    Returns the number of knife-edge mask flips that were tolerated (normally 0).
    """
    T, W = o.T, o.W
    tolerated = 0
    if teacher_forced:
        eng.upload(prev[0], prev[1], prev[2], prev[3])
        if o.tempered:
            eng.set_adapt_time(prev[4])
    labels = rec["labels"]
    tt = np.arange(T)[:, None]
    nsp = getattr(o, "nsplits", 2)
    if eng.nsplits != nsp:
        eng.set_nsplits(nsp)
    for sp in range(nsp):
        keep = eng.stretch_split(sp, labels, rec[f"rint{sp}"], rec[f"u_zz{sp}"], rec[f"u_acc{sp}"])
        ref = rec[f"keep{sp}"]
        bad = keep != ref
        if bad.any():
            with np.errstate(divide="ignore"):
                ke = knife_edge(rec[f"lnpdiff{sp}"], np.log(rec[f"u_acc{sp}"]))
            assert ke[bad].all(), f"accept mask differs off the knife edge (split {sp}): {int(bad.sum())} walkers"
            tolerated += int(bad.sum())
    x, L, P, _ = eng.download()
    if tolerated == 0:
        # positions: accepted rows are q = c - (c - s) zz computed without FMA -> bit-exact
        xs = rec[f"x_after{nsp - 1}"]
        assert np.array_equal(x, xs), f"x after stretch: max abs diff {np.abs(x - xs).max()}"
        assert np.array_equal(P, rec["P_stretch"]), "log-prior after stretch"
        rel = tol.check_logl(L, rec["L_stretch"], tol.RTOL_L if rtol_l is None else rtol_l, "log-like after stretch")
        if stats is not None:
            stats["max_rel_L"] = max(stats.get("max_rel_L", 0.0), rel)
    if o.tempered and T > 1:
        if teacher_forced and tolerated == 0:
            # swap decisions depend on L to the last bit: teacher-force the oracle's L
            eng.upload(rec[f"x_after{nsp - 1}"], rec["L_stretch"], rec["P_stretch"], prev[3])
            eng.set_adapt_time(prev[4])
        sel, swaps = eng.pt_sweep(rec["iperm"], rec["i1perm"], rec["u_swap"], adapt=True)
        bad = sel != rec["sel"]
        if bad.any():
            # recompute paccept for the knife-edge test
            assert False, f"swap mask differs in {int(bad.sum())} places"
        assert np.array_equal(swaps, rec["swaps_accepted"])
        x, L, P, betas = eng.download()
        if tolerated == 0:
            assert np.array_equal(x, rec["x"]), "x after PT"
            assert np.array_equal(P, rec["P"]), "log-prior after PT"
            if teacher_forced:
                assert np.array_equal(L, rec["L"]), "log-like after PT (teacher-forced: pure permutation)"
            else:
                tol.check_logl(L, rec["L"], tol.RTOL_L if rtol_l is None else rtol_l, "log-like after PT")
        np.testing.assert_allclose(betas, rec["betas_after"], rtol=1e-13, atol=0)
    return tolerated


def run_parity(o, eng, n_iters, teacher_forced=True, stats=None):
    """Run the oracle n_iters iterations and replay each on the engine."""
    tolerated = 0
    if not teacher_forced:
        eng.upload(o.x, o.L, o.P, o.betas)
    for _ in range(n_iters):
        prev = (o.x.copy(), o.L.copy(), o.P.copy(), None if o.betas is None else o.betas.copy(), o.time)
        o.iteration()
        tolerated += check_iteration(eng, o, o.trace[-1], prev, teacher_forced=teacher_forced, stats=stats)
        o.trace.clear()
    return tolerated
