"""CPU oracle for Eryn's reversible-jump leaf-packing path.

TEST INFRASTRUCTURE ONLY - the checker the HIP path is compared against, never shipped or measured as the product
(see oracle/eryn_oracle.py).  A flat NumPy restatement of, per sampler iteration (ensemble.py:963-1024):

  * the in-model ``GaussianMove`` on the packed active leaves of every branch   moves/mh.py:56-193, gaussian.py:68-270
  * or, as the in-model move, the red / blue ``StretchMove`` over EVERY branch and leaf slot of a walker (SURVEY 8 row a4's loop
    over branches: one complement walker per branch, one stretch factor per walker)   moves/stretch.py:55-231, red_blue.py:103-330
  * ``EnsembleSampler.compute_log_prior`` / ``compute_log_like`` with ``inds``    ensemble.py:1127-1217, 1219-1545
  * ``Move.update`` incl. ``inds``                                               moves/move.py:472-703
  * ``TemperatureControl.temper_comps`` (swaps carry every branch's leaves)       moves/tempering.py:351-649
  * ``DistributionGenerateRJ`` birth / death of one leaf per walker + edge factors
                                                                                 moves/distgenrj.py:35-222, rj.py:145-388
  * ``temper_comps(adapt=False)`` after the RJ move                               rj.py:381-382

Parity is PINNED: tests/golden/make_golden_rj.py records the reference's states on three scenarios and
tests/test_oracle_golden_rj.py requires this module to reproduce every one of them bit for bit from the two seeds
(R = the sampler-owned RandomState, G = the global np.random stream), which also pins the order of every draw.

The model is the reference tests' own (tests/test_eryn.py:38-92): branches of Gaussian pulses ``a exp(-(t-b)^2 / 2c^2)``
and sine waves ``a sin(2 pi b t + c)`` summed into a template, ``logL = -1/2 sum(((template - y) / sigma)^2)``.
All file:line citations are relative to /root/reference/src/eryn unless they name tests/.
"""
import numpy as np

from . import eryn_oracle as base

KIND_PULSE, KIND_SINE = 0, 1


class Branch:
    """One model type: leaf kind, box prior per leaf parameter, leaf budget."""

    def __init__(self, name, kind, box, nleaves_max, nleaves_min=0, cov=None):
        self.name, self.kind = name, int(kind)
        self.lo = np.array([b[0] for b in box], dtype=np.float64)
        self.hi = np.array([b[1] for b in box], dtype=np.float64)
        self.ndim = len(box)
        self.nleaves_max, self.nleaves_min = int(nleaves_max), int(nleaves_min)
        self.cov = None if cov is None else np.asarray(cov, dtype=np.float64)      # in-model proposal covariance
        self.logpdf_vals = np.log(1 / (self.hi - self.lo))                        # prior.py:28-41

    def leaf_logpdf(self, x):
        """x[N, ndim] -> sum over the parameters, accumulated sequentially from 0.0 (prior.py:364-383, 80-88)."""
        out = np.zeros(x.shape[0])
        for d in range(self.ndim):
            xd = x[:, d]
            t = np.zeros_like(xd)
            t[(xd >= self.lo[d]) & (xd <= self.hi[d])] = self.logpdf_vals[d]
            t[(xd < self.lo[d]) | (xd > self.hi[d])] = -np.inf
            out += t
        return out

    def rvs(self, n, G):
        """``ProbDistContainer.rvs`` (prior.py:432-497): one ``rand(n)`` of the GLOBAL stream per parameter, in order."""
        out = np.zeros((n, self.ndim))
        for d in range(self.ndim):
            out[:, d] = G.rand(n) * (self.hi[d] - self.lo[d]) + self.lo[d]          # prior.py:60-66
        return out


def compute_log_prior(x, inds, branches):
    """ensemble.py:1189-1210: per branch the prior of EVERY leaf slot, inactive slots zeroed, summed over the leaf axis
    with ``ndarray.sum(axis=-1)`` (NumPy's pairwise order), branches accumulated in order."""
    first = branches[0].name
    T, W = x[first].shape[:2]
    out = np.zeros((T, W))
    for br in branches:
        v = br.leaf_logpdf(x[br.name].reshape(-1, br.ndim)).reshape(T, W, br.nleaves_max)
        v[~inds[br.name]] = 0.0
        out += v.sum(axis=-1)
    if np.any(np.isnan(out)):
        raise ValueError("The prior function is returning Nan.")
    return out


def template_log_like(x, inds, branches, t, y, sigma):
    """The reference tests' likelihood for every walker at once.  Leaf by leaf in slot order, pulses before sines
    (tests/test_eryn.py:43-47,72-92); a walker's inactive slots add nothing, which leaves each walker's template
    bit-identical to the per-walker loop of the reference (x + 0.0 == x)."""
    first = branches[0].name
    T, W = x[first].shape[:2]
    tmpl = np.zeros((T, W, t.shape[0]))
    for br in branches:
        xb, ib = x[br.name], inds[br.name]
        sub = np.zeros((T, W, t.shape[0]))          # every branch sums its own leaves first (combine_gaussians / combine_sine),
        for n in range(br.nleaves_max):             # then the branch template is added to the total (tests/test_eryn.py:84-89)
            a, b, c = (xb[:, :, n, k][:, :, None] for k in range(3))
            with np.errstate(all="ignore"):
                if br.kind == KIND_PULSE:
                    # the reference squares the SCALAR c: np.float64.__pow__ goes through libm pow(), which differs from
                    # c * c (and from the array power loops) in the last bit in ~0.1 % of the cases - same call here
                    c2 = np.array([v ** 2 for v in c.ravel()]).reshape(c.shape)
                    f = a * np.exp(-((t - b) ** 2) / (2 * c2))                     # tests/test_eryn.py:38-40
                else:
                    f = a * np.sin(2 * np.pi * b * t + c)                          # tests/test_eryn.py:67-69
            sub += np.where(ib[:, :, n][:, :, None], f, 0.0)
        tmpl += sub
    return -0.5 * np.sum(((tmpl - y) / sigma) ** 2, axis=-1)


def lorentz_chirp_log_like(x_list, t, y, sigma):
    """A user likelihood that is NOT the kernel's template model (round 6: reversible jump with a host-callable likelihood), in the
    reference's calling convention for several branches without vectorisation (ensemble.py:1420-1470): ``x_list[b]`` holds the
    active leaves of branch b of ONE walker, ``[nleaves, 3]``, or None.  Branch 0: Lorentzian lines a / (1 + ((t - b) / c)^2);
    branch 1: chirps a sin(2 pi b t + c t^2).  Gaussian noise of width sigma."""
    lines, chirps = x_list
    tm = np.zeros_like(t)
    if lines is not None:
        for a, b, c in np.atleast_2d(lines):
            tm = tm + a / (1.0 + ((t - b) / c) ** 2)
    if chirps is not None:
        for a, b, c in np.atleast_2d(chirps):
            tm = tm + a * np.sin(2 * np.pi * b * t + c * t ** 2)
    return -0.5 * np.sum(((tm - y) / sigma) ** 2)


def ramp_burst_log_like(x_list, t, y, sigma):
    """A user likelihood over branches of DIFFERENT leaf widths (round 6: hens_rj_set_model_general), same calling convention as
    lorentz_chirp_log_like.  Branch 0, two parameters per leaf: ramps a + b t; branch 1, four: bursts
    a exp(-((t - t0) / w)^2) cos(2 pi f (t - t0)).  Gaussian noise of width sigma."""
    ramps, bursts = x_list
    tm = np.zeros_like(t)
    if ramps is not None:
        for a, b in np.atleast_2d(ramps):
            tm = tm + (a + b * t)
    if bursts is not None:
        for a, t0, w, f in np.atleast_2d(bursts):
            tm = tm + a * np.exp(-(((t - t0) / w) ** 2)) * np.cos(2 * np.pi * f * (t - t0))
    return -0.5 * np.sum(((tm - y) / sigma) ** 2)


def offset_log_like(offs, t, y, sigma):
    """... and ONE branch of one-parameter leaves: constant offsets (a leaf is a number).  With one model type the reference hands the
    function that branch's leaves directly, not a list over the branches (ensemble.py:1466-1467)."""
    tm = np.zeros_like(t)
    for (a,) in np.asarray(offs).reshape(-1, 1):
        tm = tm + a
    return -0.5 * np.sum(((tm - y) / sigma) ** 2)


def callable_log_like(like_fn, x, inds, evaluated, args):
    """What ensemble.py:1306-1334, 1420-1480 hands a non-vectorised user function: per evaluated walker (group) the list over the
    branches of that walker's active leaves ``coords[inds]`` - in slot order - or None for a branch without one."""
    first = next(iter(x))
    T, W = x[first].shape[:2]
    out = np.zeros((T, W))
    for tt in range(T):
        for w in range(W):
            if not evaluated[tt, w]:
                continue
            arg = []
            for name in x:
                m = inds[name][tt, w]
                arg.append(x[name][tt, w][m] if m.any() else None)
            out[tt, w] = like_fn(arg[0] if len(arg) == 1 else arg, *args)        # (one model type: taken out of the list, :1466-1467)
    return out


def compute_log_like(x, inds, logp, branches, t, y, sigma, fill=-1e300, like_fn=None):
    """ensemble.py:1219-1545 with ``inds``: walkers with an infinite prior are not evaluated, walkers without any active
    leaf get ``fill_zero_leaves_val``; everything that is not evaluated is -1e300.  ``like_fn``: a user function in the
    reference's per-walker calling convention (args = [t, y, sigma]) instead of the template model."""
    for br in branches:
        xa = x[br.name][inds[br.name]]
        if np.any(np.isinf(xa)):
            raise ValueError("At least one parameter value was infinite")          # :1258-1262
        if np.any(np.isnan(xa)):
            raise ValueError("At least one parameter value was NaN")
    if np.all(np.isinf(logp)):                                                     # :1272-1276
        return np.full_like(logp, -1e300)
    bad = np.isinf(logp)
    any_leaf = np.zeros(logp.shape, dtype=bool)
    for br in branches:
        any_leaf |= inds[br.name].any(axis=-1)
    evaluated = any_leaf & ~bad                                                    # groups_from_inds on inds_copy (:1278-1306)
    ll = np.full(logp.shape, -1e300)
    if like_fn is not None:
        vals = callable_log_like(like_fn, {br.name: x[br.name] for br in branches}, inds, evaluated, [t, y, sigma])
    else:
        vals = template_log_like(x, inds, branches, t, y, sigma)
    ll[evaluated] = vals[evaluated]
    ll[~evaluated] = fill                                                          # :1513
    if np.any(np.isnan(ll)):
        raise ValueError("The likelihood function is returning Nan.")
    return ll


def fix_logp_gibbs(logp, inds, run_names):
    """Move.fix_logp_gibbs (move.py:368-402): a walker whose branches UNDER PROPOSAL hold no leaf gets log-prior -inf if
    another branch still has leaves ("no use in running because no change": such a proposal is never accepted - in
    particular the death of a branch's last leaf while the other branch is populated), and 0.0 if the model is empty."""
    here = np.zeros(logp.shape, dtype=int)
    total = np.zeros(logp.shape, dtype=int)
    for name, iv in inds.items():
        n = iv.sum(axis=-1)
        total += n
        if name in run_names:
            here += n
    logp[(total != 0) & (here == 0)] = -np.inf
    logp[(total == 0) & (here == 0)] = 0.0


def update(st, q, new_inds, logl, logp, accepted):
    """Move.update (move.py:472-703) for the whole ensemble: log-like, log-prior (inf -> 0), inds, coordinates."""
    new_logp = logp.copy()
    new_logp[np.isinf(new_logp)] = 0.0                                             # :523-526
    st.L = logl * accepted + st.L * (~accepted)
    st.P = new_logp * accepted + st.P * (~accepted)
    for name in st.x:
        st.inds[name] = new_inds[name] * accepted[:, :, None] + st.inds[name] * (~accepted[:, :, None])
        st.x[name][accepted] = q[name][accepted]                                   # :659-682 (copy, then fill)


class RJState:
    def __init__(self, x, inds, L, P, betas):
        self.x = {k: np.array(v, dtype=np.float64, copy=True) for k, v in x.items()}
        self.inds = {k: np.array(v, dtype=bool, copy=True) for k, v in inds.items()}
        self.L, self.P = np.array(L, copy=True), np.array(P, copy=True)
        self.betas = np.array(betas, dtype=np.float64, copy=True)


class OracleRJSampler:
    """One in-model GaussianMove + one RJ move per iteration, driven by the two reference streams."""

    def __init__(self, branches, x0, inds0, t, y, sigma, R, G, betas, adaptive=True, adaptation_lag=10000,
                 adaptation_time=100, stop_adaptation=-1, fill=-1e300, record=False, schedule="separate_branches",
                 in_model="gaussian", a=2.0, like_fn=None):
        # schedule: the sampler's ``rj_moves`` string (ensemble.py:434-480) - "separate_branches": one DistributionGenerateRJ per
        # branch, one of them chosen per iteration; "iterate_branches": ONE move that walks through every branch in turn;
        # "together": ONE move that proposes a birth or death in every branch of a walker at once (ensemble.py:414-432)
        # "none": no reversible-jump move at all (EnsembleSampler without rj_moves: the leaf masks never change)
        if schedule not in ("separate_branches", "iterate_branches", "together", "none"):
            raise ValueError("rj_moves must be 'together', 'iterate_branches', or 'separate_branches'")
        if in_model not in ("gaussian", "stretch"):
            raise ValueError("in_model must be 'gaussian' or 'stretch'")
        self.schedule, self.in_model, self.a = schedule, in_model, float(a)
        self.branches = list(branches)
        self.t, self.y, self.sigma = np.asarray(t, dtype=np.float64), np.asarray(y, dtype=np.float64), float(sigma)
        self.R, self.G = R, G
        self.like_fn = like_fn
        P = compute_log_prior(x0, inds0, self.branches)
        L = compute_log_like(x0, inds0, P, self.branches, self.t, self.y, self.sigma, fill, like_fn=like_fn)
        self.st = RJState(x0, inds0, L, P, betas)
        first = self.branches[0].name
        self.T, self.W = self.st.x[first].shape[:2]
        self.adaptive, self.lag, self.nu, self.stop = adaptive, adaptation_lag, adaptation_time, stop_adaptation
        self.time, self.fill = 0, fill
        self.swaps_accepted = np.zeros(max(self.T - 1, 0))
        self.mh_accepted = np.zeros((self.T, self.W))
        self.rj_accepted = [np.zeros((self.T, self.W)) for _ in self.branches]
        self.rj_num_proposals = [0 for _ in self.branches]
        self.record, self.trace = record, []

    # ---- draw sources: the reference's two streams, call for call in the reference's order.  A replay of the device's
    # ---- production draws overrides these and nothing else (tests/test_hip_rj.py: hens_rj_step against this oracle) ------
    def _draw_move_choice(self):
        self.R.choice(1, p=np.ones(1))                                             # ensemble.py:971 (one move: still a draw)

    def _draw_steps(self, b, n):
        return self.R.multivariate_normal(np.zeros(b.ndim), b.cov, size=n)         # gaussian.py:265-268 (factor None)

    def _draw_accept(self, which):
        return self.R.rand(self.T, self.W)                                         # mh.py:157 / rj.py:332

    def _draw_split_labels(self):
        return base.split_labels(self.T, self.W, self.G)                           # red_blue.py:119-124 (shuffles of the GLOBAL stream)

    def _draw_rint(self, bi, split, Ns, Nc):
        return self.R.randint(Nc, size=(self.T, Ns))                               # stretch.py:93-99, once per branch

    def _draw_zz(self, split, Ns):
        return self.R.rand(self.T, Ns)                                             # stretch.py:129-132, first branch only

    def _draw_accept_split(self, split, Ns):
        return self.R.rand(self.T, Ns)                                             # red_blue.py:294

    def _draw_branch(self, nb):
        return int(self.R.choice(nb, p=np.full(nb, 1.0 / nb)))                     # ensemble.py:988-990, separate_branches

    def _draw_coin(self, shape):
        return self.R.choice([-1, +1], size=shape)                                 # distgenrj.py:63-66

    def _draw_leaf(self, tt, w, candidates):
        return self.R.choice(candidates)                                           # distgenrj.py:97-112

    def _draw_birth(self, b, bt, bw):
        return b.rvs(len(bt), self.G)                                              # generate_dist.rvs (prior.py:60-66), (t, w) order

    def _draw_pair(self, j, W):
        return self.G.permutation(W), self.G.permutation(W), self.G.uniform(size=W)    # tempering.py:526-535

    # ---- shared pieces ------------------------------------------------------------------------------------------
    def _logP(self, logl, logp):
        return base.tempered_log_posterior(logl, logp, self.st.betas)

    def _accept(self, factors, logl, logp, u_acc):
        logP = self._logP(logl, logp)
        prev = self._logP(self.st.L, self.st.P)
        with np.errstate(invalid="ignore"):
            lnpdiff = factors + logP - prev                                        # mh.py:155, rj.py:330
        with np.errstate(divide="ignore"):
            return lnpdiff > np.log(u_acc), lnpdiff                                # mh.py:157, rj.py:332

    def _pt(self, adapt, rec):
        """temper_comps (tempering.py:598-649): hot -> cold cascade over every branch's leaves, then adaptation."""
        T, W, st = self.T, self.W, self.st
        if T < 2:
            return
        iperm = np.empty((T - 1, W), dtype=np.int64)
        i1perm = np.empty((T - 1, W), dtype=np.int64)
        u = np.empty((T - 1, W))
        sel_all = np.zeros((T - 1, W), dtype=bool)
        arrays = [st.L, st.P] + [st.x[b.name] for b in self.branches] + [st.inds[b.name] for b in self.branches]
        for j, i in enumerate(range(T - 1, 0, -1)):                                # tempering.py:515-559
            dbeta = st.betas[i - 1] - st.betas[i]
            iperm[j], i1perm[j], u[j] = self._draw_pair(j, W)
            with np.errstate(divide="ignore"):
                sel = dbeta * (st.L[i, iperm[j]] - st.L[i - 1, i1perm[j]]) > np.log(u[j])
            sel_all[j] = sel
            self.swaps_accepted[i - 1] = np.sum(sel)
            a_, b_ = iperm[j][sel], i1perm[j][sel]
            for arr in arrays:                                                     # tempering.py:376-480
                tmp = arr[i, a_].copy()
                arr[i, a_] = arr[i - 1, b_]
                arr[i - 1, b_] = tmp
        if adapt and self.adaptive:                                                # tempering.py:632-633, 585-596
            if self.stop < 0 or self.time < self.stop:
                st.betas = base.adapt_ladder(st.betas, self.swaps_accepted, W, self.time, self.lag, self.nu)
            self.time += 1
        if rec is not None:
            rec.update(iperm=iperm, i1perm=i1perm, u_swap=u, sel=sel_all, swaps=self.swaps_accepted.copy(),
                       betas_after=st.betas.copy())

    def _snapshot(self, rec, prefix):
        st = self.st
        for b in self.branches:
            rec[f"{prefix}x_{b.name}"], rec[f"{prefix}inds_{b.name}"] = st.x[b.name].copy(), st.inds[b.name].copy()
        rec[f"{prefix}L"], rec[f"{prefix}P"] = st.L.copy(), st.P.copy()

    # ---- in-model Gaussian move on the packed leaves (mh.py:56-193, gaussian.py:68-115, 260-270) -----------------------
    def mh_move(self, rec=None):
        st = self.st
        self._draw_move_choice()
        q, steps = {}, {}
        for b in self.branches:
            inds_here = np.where(st.inds[b.name])
            x0 = st.x[b.name][inds_here]
            step = self._draw_steps(b, len(x0))
            q[b.name] = st.x[b.name].copy()
            q[b.name][inds_here] = x0 + 1.0 * step
            steps[b.name] = step
        logp = compute_log_prior(q, st.inds, self.branches)                        # mh.py:120
        fix_logp_gibbs(logp, st.inds, [b.name for b in self.branches])             # mh.py:122-124 (every branch runs)
        logl = compute_log_like(q, st.inds, logp, self.branches, self.t, self.y, self.sigma, self.fill, like_fn=self.like_fn)
        u_acc = self._draw_accept("mh")
        accepted, lnpdiff = self._accept(np.zeros((self.T, self.W)), logl, logp, u_acc)
        if rec is not None:
            self._snapshot(rec, "pre_")
            rec.update(mh_steps=steps, mh_q={k: v.copy() for k, v in q.items()}, mh_logp=logp, mh_logl=logl,
                       mh_u_acc=u_acc, mh_lnpdiff=lnpdiff, mh_accepted=accepted, betas_before=st.betas.copy(),
                       time_before=self.time)
        update(st, q, st.inds, logl, logp, accepted)
        self.mh_accepted += accepted
        if rec is not None:
            self._snapshot(rec, "mhupd_")
        self._pt(True, rec)
        return accepted

    # ---- in-model red / blue stretch move over every branch and leaf slot (red_blue.py:103-330, stretch.py:55-231) -------------
    def stretch_move(self, rec=None):
        """``RedBlueMove.propose`` with ``StretchMove.get_proposal`` on a state of several branches and leaves, no Gibbs sampling
        (``inds_run`` None everywhere: ``gibbs_ndim`` equals the full dimension, ``adjust_factors`` changes nothing, stretch.py:
        225-229).  Per half: for every branch IN ORDER one ``randint`` of R picks each moving walker's complement walker - a
        different one per branch (``choose_c_vals``, stretch.py:205) - and right behind the first branch's one ``rand`` of R gives
        the walker's stretch factor, shared by all branches (stretch.py:128-132: ``branch_i == 0``); every leaf slot moves, active
        or not (the masks enter the prior and the likelihood only); factors = (sum over branches of nleaves_max * ndim - 1) log zz
        (stretch.py:222-223); the masks of the moving walkers stay (red_blue.py:158-165); all slots of an accepted walker are
        replaced (move.py:659-682)."""
        st, T, W = self.st, self.T, self.W
        self._draw_move_choice()
        ndim_total = sum(b.nleaves_max * b.ndim for b in self.branches)
        if W < 2 * ndim_total:                                                     # red_blue.py:103-114
            raise RuntimeError("It is unadvisable to use a red-blue move with fewer walkers than twice the number of dimensions.")
        labels = self._draw_split_labels()
        accepted = np.zeros((T, W), dtype=bool)
        tt = np.arange(T)[:, None]
        names = [b.name for b in self.branches]
        if rec is not None:
            self._snapshot(rec, "pre_")
            rec.update(st_labels=labels.copy(), betas_before=st.betas.copy(), time_before=self.time)
        for split in range(2):                                                     # red_blue.py:148
            S, C = base.split_index_lists(labels, split, 2)
            Ns, Nc = S.shape[1], C.shape[1]
            q, rints, zz, u_zz = {}, [], None, None
            for bi, b in enumerate(self.branches):                                 # stretch.py:187-218
                rint = self._draw_rint(bi, split, Ns, Nc)
                if bi == 0:
                    u_zz = self._draw_zz(split, Ns)
                    zz = ((self.a - 1.0) * u_zz + 1) ** 2.0 / self.a
                s = st.x[b.name][tt, S]                                            # [T, Ns, nleaves_max, ndim]
                c = st.x[b.name][tt, C[tt, rint]]
                q[b.name] = c - (c - s) * zz[:, :, None, None]                     # stretch.py:141-145
                rints.append(rint)
            factors = (ndim_total - 1.0) * np.log(zz)                              # stretch.py:223
            new_inds = {n: st.inds[n][tt, S] for n in names}
            logp = compute_log_prior(q, new_inds, self.branches)                   # red_blue.py:260-266
            fix_logp_gibbs(logp, new_inds, names)                                  # red_blue.py:268
            logl = compute_log_like(q, new_inds, logp, self.branches, self.t, self.y, self.sigma, self.fill, like_fn=self.like_fn)
            u_acc = self._draw_accept_split(split, Ns)
            prevL, prevP = st.L[tt, S], st.P[tt, S]
            logP = base.tempered_log_posterior(logl, logp, st.betas)
            prev = base.tempered_log_posterior(prevL, prevP, st.betas)
            with np.errstate(invalid="ignore"):
                lnpdiff = factors + logP - prev                                    # red_blue.py:292
            with np.errstate(divide="ignore"):
                keep = lnpdiff > np.log(u_acc)                                     # red_blue.py:294
            new_logp = logp.copy()
            new_logp[np.isinf(new_logp)] = 0.0                                     # move.py:523-526
            st.L[tt, S] = logl * keep + prevL * (~keep)
            st.P[tt, S] = new_logp * keep + prevP * (~keep)
            for n in names:                                                        # move.py:659-682: every slot of an accepted walker
                xs = st.x[n][tt, S].copy()
                xs[keep] = q[n][keep]
                st.x[n][tt, S] = xs
            accepted[tt, S] = keep
            if rec is not None:
                rec.update({f"st_rint{split}": np.stack(rints), f"st_u_zz{split}": u_zz, f"st_u_acc{split}": u_acc,
                            f"st_q{split}": {k: v.copy() for k, v in q.items()}, f"st_logp{split}": logp, f"st_logl{split}": logl,
                            f"st_factors{split}": factors, f"st_lnpdiff{split}": lnpdiff, f"st_keep{split}": keep,
                            f"st_S{split}": S, f"st_C{split}": C})
                self._snapshot(rec, f"stupd{split}_")
        self.mh_accepted += accepted
        if rec is not None:
            rec["mh_accepted"] = accepted
            self._snapshot(rec, "mhupd_")
        self._pt(True, rec)
        return accepted

    # ---- reversible jump: one leaf born or killed per walker in ONE branch (distgenrj.py:35-222, rj.py:145-388) ------
    def rj_move(self, rec=None):
        nb = len(self.branches)
        if self.schedule == "iterate_branches":
            # one move object (ensemble.py:434-451: the choice among ONE move still draws), the Gibbs iterator hands it the
            # branches one at a time (rj.py:169-171): propose / accept / update per branch, `accepted` overwritten each time,
            # then ONE sweep of swaps without adaptation (rj.py:381-382)
            self._draw_branch(1)
            accepted, sub = None, []
            for bi in range(nb):
                r = {} if rec is not None else None
                accepted = self._rj_branch(bi, r)
                sub.append(r)
            rec2 = {} if rec is not None else None
            self._pt(False, rec2)
            if rec is not None:
                rec.update(rj_sub=sub, rj_accepted=accepted)
                rec.update({f"rj_{k}": v for k, v in rec2.items()})
            self.rj_accepted[0] += accepted                                        # rj.py:385-386: the LAST branch's mask
            self.rj_num_proposals[0] += 1
            return 0, accepted
        if self.schedule == "together":
            self._draw_branch(1)                                                   # (the choice among ONE move still draws)
            accepted = self._rj_propose(list(range(nb)), rec)
            rec2 = {} if rec is not None else None
            self._pt(False, rec2)
            if rec is not None:
                rec.update({f"rj_{k}": v for k, v in rec2.items()})
            self.rj_accepted[0] += accepted
            self.rj_num_proposals[0] += 1
            return 0, accepted
        bi = self._draw_branch(nb)
        accepted = self._rj_branch(bi, rec)
        rec2 = {} if rec is not None else None
        self._pt(False, rec2)                                                      # rj.py:381-382
        if rec is not None:
            rec.update({f"rj_{k}": v for k, v in rec2.items()})
        self.rj_accepted[bi] += accepted
        self.rj_num_proposals[bi] += 1
        return bi, accepted

    def _rj_branch(self, bi, rec=None):
        """Birth / death on ONE branch: proposal, factors, prior + likelihood, accept, update (rj.py:169-352)."""
        return self._rj_propose([bi], rec)

    def _rj_propose(self, bis, rec=None):
        """One birth / death proposal over the branches ``bis`` (one: "separate_branches" / "iterate_branches"; all of them:
        "together") - DistributionGenerateRJ.get_proposal (distgenrj.py:150-222) inside ReversibleJumpMove.propose
        (rj.py:169-352): every branch's coin and leaf choices first, then deaths / births and their factors branch by branch,
        edge factors, prior + likelihood, ONE accept test, update."""
        st, T, W = self.st, self.T, self.W
        q = {k: v.copy() for k, v in st.x.items()}
        new_inds = {k: v.copy() for k, v in st.inds.items()}
        factors = np.zeros((T, W))
        changes, leaves, births = {}, {}, {}
        for bi in bis:                                                             # distgenrj.py:166-180 (first loop: R only)
            b = self.branches[bi]
            self._bi = bi
            inds = st.inds[b.name]
            nleaves = inds.sum(axis=-1)
            change = np.zeros((T, W), dtype=np.int64)
            leaf = np.full((T, W), -1, dtype=np.int64)
            if b.nleaves_min != b.nleaves_max:
                change = self._draw_coin(nleaves.shape)
                change = (change * ((nleaves != b.nleaves_min) & (nleaves != b.nleaves_max))
                          + (+1) * (nleaves == b.nleaves_min) + (-1) * (nleaves == b.nleaves_max))   # :69-73
                for tt in range(T):                                                # :85-121, one draw per walker, in order
                    for w in range(W):
                        if change[tt, w] == +1:
                            leaf[tt, w] = self._draw_leaf(tt, w, np.where(~inds[tt, w])[0])
                        elif change[tt, w] == -1:
                            leaf[tt, w] = self._draw_leaf(tt, w, np.where(inds[tt, w])[0])
            changes[bi], leaves[bi] = change, leaf
        for bi in bis:                                                             # :183-220 (second loop: G for the births)
            b = self.branches[bi]
            self._bi = bi
            change, leaf = changes[bi], leaves[bi]
            births[bi] = np.zeros((0, b.ndim))
            if b.nleaves_min != b.nleaves_max:
                dt, dw = np.where(change == -1)                                    # deaths first (:188-197)
                dl = leaf[dt, dw]
                new_inds[b.name][dt, dw, dl] = False
                np.add.at(factors, (dt, dw), +1 * b.leaf_logpdf(q[b.name][dt, dw, dl]))
                bt, bw = np.where(change == +1)                                    # births (:199-214)
                bl = leaf[bt, bw]
                new_inds[b.name][bt, bw, bl] = True
                births[bi] = self._draw_birth(b, bt, bw)
                q[b.name][bt, bw, bl] = births[bi]
                np.add.at(factors, (bt, bw), -1 * b.leaf_logpdf(q[b.name][bt, bw, bl]))
        # edge factors (rj.py:236-270): one array, every branch under proposal adds to it in branch order
        edge = np.zeros((T, W))
        for bi in bis:
            b = self.branches[bi]
            if not (b.nleaves_min == b.nleaves_max or b.nleaves_min + 1 == b.nleaves_max):
                nleaves = st.inds[b.name].sum(axis=-1)
                new_nl = new_inds[b.name].sum(axis=-1)
                edge[nleaves == b.nleaves_min] += np.log(1 / 2.0)
                edge[nleaves == b.nleaves_max] += np.log(1 / 2.0)
                edge[new_nl == b.nleaves_min] -= np.log(1 / 2.0)
                edge[new_nl == b.nleaves_max] -= np.log(1 / 2.0)
        factors += edge
        logp = compute_log_prior(q, new_inds, self.branches)                       # rj.py:300
        fix_logp_gibbs(logp, new_inds, [self.branches[bi].name for bi in bis])     # rj.py:302 (the branches under proposal)
        logl = compute_log_like(q, new_inds, logp, self.branches, self.t, self.y, self.sigma, self.fill, like_fn=self.like_fn)
        self._bi = bis[0] if len(bis) == 1 else len(self.branches)
        u_acc = self._draw_accept("rj")
        accepted, lnpdiff = self._accept(factors, logl, logp, u_acc)
        if rec is not None:
            self._snapshot(rec, "rjpre_")
            if len(bis) == 1:
                rec.update(rj_branch=bis[0], rj_change=changes[bis[0]].copy(), rj_leaf=leaves[bis[0]].copy(),
                           rj_birth=births[bis[0]].copy())
            else:
                rec.update(rj_branches=list(bis), rj_change_all=[changes[bi].copy() for bi in bis],
                           rj_leaf_all=[leaves[bi].copy() for bi in bis], rj_birth_all=[births[bi].copy() for bi in bis])
            rec.update(rj_factors=factors.copy(), rj_logp=logp, rj_logl=logl, rj_u_acc=u_acc, rj_lnpdiff=lnpdiff,
                       rj_accepted=accepted, rj_q={k: v.copy() for k, v in q.items()},
                       rj_new_inds={k: v.copy() for k, v in new_inds.items()})
        update(st, q, new_inds, logl, logp, accepted)
        if rec is not None:
            self._snapshot(rec, "rjupd_")
        return accepted

    def iteration(self):
        rec = {} if self.record else None
        acc = self.mh_move(rec) if self.in_model == "gaussian" else self.stretch_move(rec)
        if rec is not None:
            self._snapshot(rec, "mh_")
        if self.schedule == "none":
            if rec is not None:
                self.trace.append(rec)
            return acc, -1, None
        bi, racc = self.rj_move(rec)
        if rec is not None:
            self._snapshot(rec, "rj_")
            self.trace.append(rec)
        return acc, bi, racc
