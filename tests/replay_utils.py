"""Replay of PRODUCTION (Philox) iterations through the CPU oracle.

``hens_step`` draws on the device; its draws are a pure function of (seed, iteration, global rung, walker) and
``hens_debug_draws`` exports them in a form that maps one-to-one onto the reference's draws:

    labels  <- which half of the position list a walker is in      (red_blue.py:119-124)
    rint    <- index of the complement walker in the ascending complement list  (stretch.py:93-99)
    u_zz, u_acc                                                    (stretch.py:129-132, red_blue.py:294)
    iperm, i1perm <- slots that cascade column c visits on rungs i and i-1   (tempering.py:526-541)
    u_swap                                                         (tempering.py:535)

``oracle_iterations`` then runs the pinned NumPy restatement (oracle/eryn_oracle.py) with exactly those draws, so
the code path the benchmark times (plan kernel, Philox cascade, folded ladder adaptation, pipeline) is held to the
same oracle as the parity API.  The conversion asserts what the reference guarantees structurally: halves of size
ceil(W/2) / floor(W/2), every complement walker in the OTHER half, permutations - and sorts the per-place draws into the
ascending order in which the reference's boolean masks enumerate a half.
"""
import numpy as np

from oracle import eryn_oracle as orc
from tests import tolerance_log as tol


def is_permutation_rows(a, W):
    return a.shape[-1] == W and np.array_equal(np.sort(a, axis=-1), np.broadcast_to(np.arange(W), a.shape))


def draws_to_reference_sets(d, T, W, nsplits):
    """The same for a red-blue move of nsplits > 2 sets (red_blue.py:41-47,119-124,148-197): positions run through the sets one
    after the other, set k holds ceil((W - k) / nsplits) walkers in ascending order, and a complement is drawn from the OTHER
    sets concatenated in set order (not their ascending merge: fixture f8_nsplits3)."""
    own, cw = d["own"].astype(np.int64), d["cw"].astype(np.int64)
    assert own.shape == (T, W) and is_permutation_rows(own, W), "own must list every walker of a rung exactly once"
    off = np.concatenate([[0], np.cumsum([(W - k + nsplits - 1) // nsplits for k in range(nsplits)])])
    assert off[-1] == W
    tt = np.arange(T)[:, None]
    labels = np.empty((T, W), dtype=np.int64)
    for k in range(nsplits):
        seg = own[:, off[k]:off[k + 1]]
        assert np.all(np.diff(seg, axis=1) > 0), "every set is listed in ascending walker order"
        labels[tt, seg] = k
    out = dict(labels=labels)
    for k in range(nsplits):
        sl = slice(off[k], off[k + 1])
        C = np.concatenate([own[:, :off[k]], own[:, off[k + 1]:]], axis=1)          # the other sets in set order
        pos = np.empty((T, W), dtype=np.int64)
        pos[tt, C] = np.arange(C.shape[1])[None, :]
        assert np.all(labels[tt, cw[:, sl]] != k), "a complement walker drawn from the moving set"
        out[f"rint{k}"] = pos[tt, cw[:, sl]]
        assert np.array_equal(C[tt, out[f"rint{k}"]], cw[:, sl])
        out[f"u_zz{k}"] = d["u_zz"][:, sl]
        out[f"u_acc{k}"] = d["u_acc"][:, sl]
    if T > 1 and "pt_slot" in d:
        slot = d["pt_slot"].astype(np.int64)
        assert slot.shape == (T, W) and is_permutation_rows(slot, W)
        out["iperm"] = slot[:0:-1].copy()
        out["i1perm"] = slot[-2::-1].copy()
        out["u_swap"] = d["u_swap"]
    return out


def draws_to_reference(d, T, W):
    """hens_debug_draws output for the whole ladder -> dict(labels, rint0/1, u_zz0/1, u_acc0/1, iperm, i1perm, u_swap)."""
    N0 = (W + 1) // 2
    own, cw = d["own"].astype(np.int64), d["cw"].astype(np.int64)
    assert own.shape == (T, W) and is_permutation_rows(own, W), "own must list every walker of a rung exactly once"
    halves = (own[:, :N0], own[:, N0:])                # by place (block labels) or ascending (k_plan)
    labels = np.empty((T, W), dtype=np.int64)
    tt = np.arange(T)[:, None]
    labels[tt, halves[0]] = 0
    labels[tt, halves[1]] = 1
    assert np.all((labels == 0).sum(axis=1) == N0)     # arange(W) % 2 shuffled: ceil(W/2) zeros (red_blue.py:120-124)
    out = dict(labels=labels)
    for sp in (0, 1):
        sl = slice(0, N0) if sp == 0 else slice(N0, W)
        # the reference enumerates both sets through boolean masks, i.e. in ascending walker order (red_blue.py:150-154,
        # 183-197): its k-th draw belongs to the k-th smallest moving walker, rint indexes the ascending complement list
        order = np.argsort(halves[sp], axis=1, kind="stable")
        C = np.sort(halves[1 - sp], axis=1)
        cwp = np.take_along_axis(cw[:, sl], order, axis=1)
        assert np.all(labels[tt, cwp] == 1 - sp), "a complement walker drawn from the moving half"
        rint = np.stack([np.searchsorted(C[t], cwp[t]) for t in range(T)])
        assert np.array_equal(C[tt, rint], cwp)
        out[f"rint{sp}"] = rint
        out[f"u_zz{sp}"] = np.take_along_axis(d["u_zz"][:, sl], order, axis=1)
        out[f"u_acc{sp}"] = np.take_along_axis(d["u_acc"][:, sl], order, axis=1)
        for k in ("u_zz", "u_acc"):
            assert np.all((out[f"{k}{sp}"] >= 0.0) & (out[f"{k}{sp}"] < 1.0))
    if T > 1 and "pt_slot" in d:
        slot = d["pt_slot"].astype(np.int64)
        assert slot.shape == (T, W) and is_permutation_rows(slot, W), "every rung's column map must be a permutation"
        out["iperm"] = slot[:0:-1].copy()              # row j: pair i = T-1-j, hot slots
        out["i1perm"] = slot[-2::-1].copy()            # cold slots of the same pairs
        out["u_swap"] = d["u_swap"]
        assert np.all((d["u_swap"] >= 0.0) & (d["u_swap"] < 1.0))
    return out


class OracleState:
    def __init__(self, x, L, P, betas, time=0):
        self.x, self.L, self.P = np.array(x, copy=True), np.array(L, copy=True), np.array(P, copy=True)
        self.betas = None if betas is None else np.array(betas, copy=True)
        self.time = int(time)
        T, W, _ = self.x.shape
        self.accepted = np.zeros((T, W))
        self.mh_accepted = np.zeros((T, W))
        self.swaps_total = np.zeros(max(T - 1, 0))
        self.swaps_last = np.zeros(max(T - 1, 0))
        self.min_margin = np.inf                        # closest accept / swap decision to its knife edge


def _margin(st, lnpdiff, logu):
    with np.errstate(invalid="ignore"):
        m = np.abs(lnpdiff - logu) / np.maximum(1.0, np.abs(lnpdiff))
    m = m[np.isfinite(m)]
    if m.size:
        st.min_margin = min(st.min_margin, float(m.min()))


def oracle_iteration(st, ref, loglike, lo, hi, a=2.0, adaptive=True, lag=10000, nu=100, stop_adaptation=-1,
                     mh=None, period=None, nsplits=2):
    """One sampler iteration on ``st`` with the given draws (ensemble.py:965-981): the stretch move's two halves
    (or one Metropolis-Hastings proposal when ``mh = (step, u_acc)``), the PT cascade, the ladder adaptation."""
    T, W, D = st.x.shape
    tt = np.arange(T)[:, None]
    if mh is not None:
        out = orc.mh_step(st.x, st.L, st.P, st.betas, mh[0], mh[1], lo, hi, loglike, period=period)
        st.mh_accepted += out["keep"]
        with np.errstate(divide="ignore"):
            _margin(st, out["lnpdiff"], np.log(mh[1]))
    else:
        for sp in range(nsplits):
            out = orc.stretch_split(st.x, st.L, st.P, st.betas, ref["labels"], sp, ref[f"rint{sp}"], ref[f"u_zz{sp}"],
                                    ref[f"u_acc{sp}"], a, lo, hi, loglike, period=period, nsplits=nsplits)
            acc = np.zeros((T, W))
            acc[tt, out["S"]] = out["keep"]
            st.accepted += acc
            with np.errstate(divide="ignore"):
                _margin(st, out["lnpdiff"], np.log(ref[f"u_acc{sp}"]))
    if st.betas is not None and T > 1:
        sel, sw = orc.pt_sweep(st.x, st.L, st.P, st.betas, ref["iperm"], ref["i1perm"], ref["u_swap"])
        st.swaps_last = sw
        st.swaps_total += sw
        if adaptive:                                                     # tempering.py:632-633
            if stop_adaptation < 0 or st.time < stop_adaptation:
                st.betas = orc.adapt_ladder(st.betas, sw, W, st.time, lag, nu)
            st.time += 1
    return st


def replay(eng_draws, st, it0, n, loglike, lo, hi, mh=False, **kw):
    """Run the oracle over iterations it0 .. it0+n-1 with the draws exported by ``eng_draws`` (a whole-ladder
    HipEnsemble).  Returns the list of per-iteration move kinds ("stretch" / "mh")."""
    T, W, _ = st.x.shape
    kinds = []
    for it in range(it0, it0 + n):
        d = eng_draws.debug_draws(it, mh=mh)
        nsp = int(kw.get("nsplits", 2))
        ref = draws_to_reference(d, T, W) if nsp == 2 else draws_to_reference_sets(d, T, W, nsp)
        if d["is_mh"]:
            oracle_iteration(st, ref, loglike, lo, hi, mh=(d["mh_step"], d["mh_u"]), **kw)
            kinds.append("mh")
        else:
            oracle_iteration(st, ref, loglike, lo, hi, **kw)
            kinds.append("stretch")
    return kinds


def assert_state_equal(st, x, L, P, betas, counters=None, mh_counters=None, rtol_l=None, what=""):
    """Bars of SURVEY 8c: positions / log-prior / counters exact, log-likelihood rtol 1e-13 (tests/tolerance_log.py records what was observed), betas rtol 1e-13."""
    assert np.array_equal(x, st.x), f"{what}: x differs from the oracle in {int((x != st.x).any(axis=-1).sum())} walkers " \
                                    f"(closest decision margin {st.min_margin:.2e})"
    assert np.array_equal(P, st.P), f"{what}: log-prior differs"
    tol.check_logl(L, st.L, tol.RTOL_L if rtol_l is None else rtol_l, what)
    if st.betas is not None:
        np.testing.assert_allclose(betas, st.betas, rtol=1e-13, atol=0, err_msg=f"{what}: betas")
    if counters is not None:
        assert np.array_equal(counters["accepted"], st.accepted), f"{what}: accept counters"
        if st.betas is not None and st.x.shape[0] > 1:
            assert np.array_equal(counters["swaps_total"], st.swaps_total), f"{what}: swap totals"
            assert np.array_equal(counters["swaps_last"], st.swaps_last), f"{what}: last sweep's swap counts"
    if mh_counters is not None:
        assert np.array_equal(mh_counters["accepted"], st.mh_accepted), f"{what}: MH accept counters"
