"""CPU oracle for Eryn's stretch-move + parallel-tempering hot path.

TEST INFRASTRUCTURE ONLY.  This module is a flat NumPy restatement of the
reference algorithm (mikekatz04/Eryn v1.2.6, 100 % Python/NumPy).  It is the
checker the HIP path is compared against; it is never the thing shipped or
measured as the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.

Parity is PINNED: ``tests/golden/make_golden.py`` imports the real reference
(``/root/reference/src``) in the build container, records every random draw
and every intermediate of the path, and ``tests/test_oracle_golden.py``
checks this restatement bit-for-bit against those committed fixtures.

All ``file:line`` citations are relative to ``/root/reference/src/eryn``.

Conventions (single branch, ``nleaves_max == 1`` squeezed away):
    x[T, W, D]  coordinates, L[T, W] log-likelihood, P[T, W] log-prior,
    betas[T]    inverse temperatures.
Two random streams drive the path, exactly as in the reference:
    R = the sampler-owned ``RandomState`` (``model.random``)
    G = the process-global ``np.random`` stream (split shuffle, PT draws)
Both are passed in as explicit ``numpy.random.RandomState`` objects
(``RandomState(seed)`` reproduces ``np.random.seed(seed)`` exactly).
"""

import numpy as np

# --------------------------------------------------------------------------
# ladder construction                                 moves/tempering.py:10-197
# --------------------------------------------------------------------------

# Temperature-step table for a 25 % swap rate on a D-dimensional Gaussian,
# D = 1..100 (numeric data originally from ptemcee; moves/tempering.py:57-160).
_TSTEP = np.array([
    25.2741, 7.0, 4.47502, 3.5236, 3.0232, 2.71225, 2.49879, 2.34226, 2.22198,
    2.12628, 2.04807, 1.98276, 1.92728, 1.87946, 1.83774, 1.80096, 1.76826,
    1.73895, 1.7125, 1.68849, 1.66657, 1.64647, 1.62795, 1.61083, 1.59494,
    1.58014, 1.56632, 1.55338, 1.54123, 1.5298, 1.51901, 1.50881, 1.49916,
    1.49, 1.4813, 1.47302, 1.46512, 1.45759, 1.45039, 1.4435, 1.4369,
    1.43056, 1.42448, 1.41864, 1.41302, 1.40761, 1.40239, 1.39736, 1.3925,
    1.38781, 1.38327, 1.37888, 1.37463, 1.37051, 1.36652, 1.36265, 1.35889,
    1.35524, 1.3517, 1.34825, 1.3449, 1.34164, 1.33847, 1.33538, 1.33236,
    1.32943, 1.32656, 1.32377, 1.32104, 1.31838, 1.31578, 1.31325, 1.31076,
    1.30834, 1.30596, 1.30364, 1.30137, 1.29915, 1.29697, 1.29484, 1.29275,
    1.29071, 1.2887, 1.28673, 1.2848, 1.28291, 1.28106, 1.27923, 1.27745,
    1.27569, 1.27397, 1.27227, 1.27061, 1.26898, 1.26737, 1.26579, 1.26424,
    1.26271, 1.26121, 1.25973,
])


def make_ladder(ndim, ntemps=None, Tmax=None):
    """Geometric beta ladder (moves/tempering.py:10-197)."""
    if type(ndim) != int or ndim < 1:
        raise ValueError("Invalid number of dimensions specified.")
    if ntemps is None and Tmax is None:
        raise ValueError("Must specify one of ``ntemps`` and ``Tmax``.")
    if Tmax is not None and Tmax <= 1:
        raise ValueError("``Tmax`` must be greater than 1.")
    if ntemps is not None and (type(ntemps) != int or ntemps < 1):
        raise ValueError("Invalid number of temperatures specified.")

    if ndim > _TSTEP.shape[0]:                       # :162-165
        tstep = 1.0 + 2.0 * np.sqrt(np.log(4.0)) / np.sqrt(ndim)
    else:
        tstep = _TSTEP[ndim - 1]

    append_inf = False
    if Tmax == np.inf:                               # :171-176
        append_inf = True
        Tmax = None
        ntemps = ntemps - 1

    if ntemps is not None:
        if Tmax is None:
            Tmax = tstep ** (ntemps - 1)             # :178-181
    else:
        if Tmax is None:
            raise ValueError("Must specify at least one of ``ntemps and finite ``Tmax``.")
        ntemps = int(np.log(Tmax) / np.log(tstep) + 2)

    betas = np.logspace(0, -np.log10(Tmax), ntemps)  # :191
    if append_inf:
        betas = np.concatenate((betas, [0]))
    return betas


# --------------------------------------------------------------------------
# model pieces
# --------------------------------------------------------------------------

def box_logpdf_vals(lo, hi):
    """Per-dimension ``log(1/(max-min))`` (prior.py:28-41)."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    return np.log(1 / (hi - lo))


def box_log_prior(q, lo, hi):
    """Independent uniform box prior, summed sequentially over dimensions.

    q[N, D] -> logp[N].  prior.py:80-88 (inclusive bounds, -inf outside) and
    prior.py:364-383 (accumulate ``prior_vals += temp`` for d = 0..D-1 from 0.0).
    """
    q = np.asarray(q)
    lvals = box_logpdf_vals(lo, hi)
    out = np.zeros(q.shape[0])
    for d in range(q.shape[1]):
        xd = q[:, d]
        t = np.zeros_like(xd)
        t[(xd >= lo[d]) & (xd <= hi[d])] = lvals[d]
        t[(xd < lo[d]) | (xd > hi[d])] = -np.inf
        out += t
    return out


def gaussian_log_like(x, mu, invcov):
    """The reference tests' Gaussian (tests/test_eryn.py:33-35), vectorised."""
    diff = x - mu
    return -0.5 * (diff * np.dot(invcov, diff.T).T).sum(axis=1)


def gaussian_diag_log_like(x, mu, invvar):
    diff = x - mu
    return -0.5 * (diff * (invvar * diff)).sum(axis=1)


def rosenbrock_log_like(x, a=1.0, b=100.0):
    """-(sum_i b (x_{i+1} - x_i^2)^2 + (a - x_i)^2).  Not in the reference;
    BASELINE config 5's stress target."""
    x0 = x[:, :-1]
    x1 = x[:, 1:]
    return -np.sum(b * (x1 - x0 * x0) ** 2 + (a - x0) ** 2, axis=1)


def compute_log_like(q, logp, loglike, fill=-1e300):
    """ensemble.py:1219-1545 on this path: skip -inf-prior walkers, fill -1e300.

    q[T, Ns, D], logp[T, Ns] -> logl[T, Ns].
    """
    T, Ns, D = q.shape
    if np.any(np.isinf(q)):
        raise ValueError("At least one parameter value was infinite")      # :1258-1262
    if np.any(np.isnan(q)):
        raise ValueError("At least one parameter value was NaN")
    if np.all(np.isinf(logp)):                                              # :1272-1276
        return np.full_like(logp, -1e300)
    valid = ~np.isinf(logp).reshape(-1)
    ll = np.full(T * Ns, -1e300)                                            # :1486
    res = np.asarray(loglike(q.reshape(-1, D)[valid]))
    ll[valid] = res
    ll[~valid] = fill                                                       # :1513
    if np.any(np.isnan(ll)):
        raise ValueError("The likelihood function is returning Nan.")      # :1541-1542
    return ll.reshape(T, Ns)


def tempered_log_posterior(logl, logp, betas):
    """moves/tempering.py:284-349 (betas=None -> logl + logp, moves/move.py:443-457)."""
    if betas is None:
        return logl + logp
    with np.errstate(invalid="ignore"):
        loglT = logl * betas[:, None]
    loglT[np.isnan(loglT)] = -np.inf
    return loglT + logp


# --------------------------------------------------------------------------
# red/blue stretch step                   moves/red_blue.py, moves/stretch.py
# --------------------------------------------------------------------------

def split_labels(T, W, G, nsplits=2, randomize=True):
    """Per-rung red/blue labels (moves/red_blue.py:119-124)."""
    labels = np.tile(np.arange(W), (T, 1)) % nsplits
    if randomize:
        for row in labels:
            G.shuffle(row)
    return labels


def split_index_lists(labels, split, nsplits=2):
    """Walker indices of the moving set S (ascending: a boolean mask enumerates it, moves/red_blue.py:150-154) and of its
    complement C: the OTHER sets, each ascending, concatenated in set order (red_blue.py:183-197 builds the list
    ``sets[:split] + sets[split+1:]``, stretch.py:199 concatenates it) - for two sets simply the other set."""
    T, W = labels.shape
    S = np.stack([np.flatnonzero(labels[t] == split) for t in range(T)])
    C = np.stack([np.concatenate([np.flatnonzero(labels[t] == j) for j in range(nsplits) if j != split]) for t in range(T)])
    return S, C


def periodic_distance(s, c, period):
    """``PeriodicContainer.distance(p1 = s, p2 = c)`` for one branch (utils/periodic.py:49-116): ``c - s``, taken the
    short way round where it exceeds half a period.  ``period``: [D], 0 = not periodic, or None."""
    diff = c - s                                                          # periodic.py:80
    if period is None:
        return diff
    idx = np.flatnonzero(np.asarray(period) > 0)
    if idx.size:
        per = np.asarray(period, dtype=np.float64)[idx]
        dp = diff[..., idx]                                               # periodic.py:93
        fix = np.abs(dp) > per / 2.0                                      # periodic.py:96-98
        new_s = -(per - s[..., idx]) * (dp < 0.0) + (per + s[..., idx]) * (dp >= 0.0)   # periodic.py:101-107
        dp[fix] = c[..., idx][fix] - new_s[fix]                           # periodic.py:110-112
        diff[..., idx] = dp
    return diff


def periodic_wrap(q, period):
    """``PeriodicContainer.wrap`` for one branch (utils/periodic.py:118-151): NumPy's remainder on the periodic
    parameters, in place."""
    if period is not None:
        idx = np.flatnonzero(np.asarray(period) > 0)
        if idx.size:
            q[..., idx] = q[..., idx] % np.asarray(period, dtype=np.float64)[idx]   # periodic.py:143-145
    return q


def stretch_split(x, L, P, betas, labels, split, rint, u_zz, u_acc, a, lo, hi,
                  loglike, fill=-1e300, period=None, nsplits=2):
    """One red/blue half-step, all rungs; mutates x, L, P in place.

    Returns a dict of intermediates (q, logp, logl, factors, lnpdiff, keep).
    SURVEY 3.2 step 3 a-j.  ``period``: periodic parameters (stretch.py:136-154), see ``periodic_distance``.
    """
    T, W, D = x.shape
    S, C = split_index_lists(labels, split, nsplits)
    Ns, Nc = S.shape[1], C.shape[1]
    tt = np.arange(T)[:, None]

    s = x[tt, S]                                   # [T, Ns, D]
    c = x[tt, C[tt, rint]]                         # stretch.py:93-100
    zz = ((a - 1.0) * u_zz + 1) ** 2.0 / a         # stretch.py:129-132
    diff = periodic_distance(s, c, period)         # stretch.py:136-143
    q = c - diff * zz[:, :, None]                  # stretch.py:145
    q = periodic_wrap(q, period)                   # stretch.py:149-154
    factors = (D - 1.0) * np.log(zz)               # stretch.py:223

    logp = box_log_prior(q.reshape(-1, D), lo, hi).reshape(T, Ns)
    if np.any(np.isnan(logp)):
        raise ValueError("The prior function is returning Nan.")            # ensemble.py:1214
    logl = compute_log_like(q, logp, loglike, fill=fill)
    logl[np.isnan(logl)] = -1e300                  # red_blue.py:279-281

    logP = tempered_log_posterior(logl, logp, betas)
    prev_logl = L[tt, S]
    prev_logp = P[tt, S]
    prev_logP = tempered_log_posterior(prev_logl, prev_logp, betas)
    with np.errstate(invalid="ignore"):
        lnpdiff = factors + logP - prev_logP       # red_blue.py:292
    with np.errstate(divide="ignore"):
        keep = lnpdiff > np.log(u_acc)             # red_blue.py:294

    # Move.update (moves/move.py:513-532, 669-682)
    new_logp = logp.copy()
    new_logp[np.isinf(new_logp)] = 0.0
    L[tt, S] = logl * keep + prev_logl * (~keep)
    P[tt, S] = new_logp * keep + prev_logp * (~keep)
    xs = s.copy()
    xs[keep] = q[keep]
    x[tt, S] = xs
    return dict(S=S, C=C, q=q, logp=logp, logl=logl, factors=factors,
                lnpdiff=lnpdiff, keep=keep, zz=zz)


# --------------------------------------------------------------------------
# Metropolis-Hastings moves             moves/mh.py, gaussian.py
# --------------------------------------------------------------------------

class GaussianProposal:
    """``GaussianMove``'s proposal function for one branch (gaussian.py:27-66, 197-270).

    ``cov`` scalar -> isotropic, 2-D -> full covariance (``rng.multivariate_normal``); a 1-D covariance
    cannot be constructed in the reference at this numpy (gaussian.py:144 raises) and is not restated.
    ``draw_step`` consumes R exactly like ``proposal_fn(coords[inds_here], random)`` and returns the
    step ``q - x0`` (zero where ``mode`` leaves a coordinate untouched), so that ``q = x0 + step``
    reproduces the reference's ``x0 + factor * scale * randn`` bit for bit.
    """

    def __init__(self, cov, mode="vector", factor=None):
        cov = np.asarray(cov, dtype=np.float64)
        if cov.ndim == 0:
            self.kind, self.scale = "iso", np.sqrt(float(cov))            # gaussian.py:60
        elif cov.ndim == 2 and cov.shape[0] == cov.shape[1]:
            self.kind, self.scale = "full", cov                           # gaussian.py:53
            if mode != "vector":
                raise ValueError("full-covariance proposals only run in 'vector' mode")   # gaussian.py:260
        else:
            raise ValueError("Invalid proposal scale dimensions")
        if factor is not None and factor < 1.0:
            raise ValueError("'factor' must be >= 1.0")                   # gaussian.py:149-150
        if mode not in ("vector", "random", "sequential"):
            raise ValueError("unrecognized mode")
        self.log_factor = None if factor is None else np.log(factor)
        self.mode = mode
        self.index = 0

    def get_factor(self, R):                                              # gaussian.py:161-164
        if self.log_factor is None:
            return 1.0
        return np.exp(R.uniform(-self.log_factor, self.log_factor))

    def draw_step(self, R, n, D):
        f = self.get_factor(R)
        if self.kind == "full":                                           # gaussian.py:265-268
            step = f * R.multivariate_normal(np.zeros(D), self.scale, size=n)
        else:                                                             # gaussian.py:166-167
            step = f * self.scale * R.randn(n, D)
        if self.mode == "vector":
            return step
        if self.mode == "random":                                         # gaussian.py:172-173
            m = R.randint(D, size=n)
        else:                                                             # gaussian.py:174-176
            m = self.index % D + np.zeros(n, dtype=int)
            self.index = (self.index + 1) % D
        out = np.zeros_like(step)
        out[np.arange(n), m] = step[np.arange(n), m]
        return out


def mh_step(x, L, P, betas, step, u_acc, lo, hi, loglike, fill=-1e300, period=None):
    """One full-ensemble Metropolis-Hastings proposal q = x + step; mutates x, L, P in place.

    ``MHMove.propose`` (mh.py:56-193) for a single branch with every leaf active: no red/blue split,
    factors = 0 (gaussian.py:131), same tempered accept test and ``Move.update`` as the stretch move.
    """
    T, W, D = x.shape
    q = periodic_wrap(x + step.reshape(T, W, D), period)                  # gaussian.py:110-115
    logp = box_log_prior(q.reshape(-1, D), lo, hi).reshape(T, W)          # mh.py:120
    if np.any(np.isnan(logp)):
        raise ValueError("The prior function is returning Nan.")
    logl = compute_log_like(q, logp, loglike, fill=fill)                  # mh.py:126-132
    logP = tempered_log_posterior(logl, logp, betas)                      # mh.py:142
    prev_logP = tempered_log_posterior(L, P, betas)                       # mh.py:146-152
    with np.errstate(invalid="ignore"):
        lnpdiff = np.zeros((T, W)) + logP - prev_logP                     # mh.py:155
    with np.errstate(divide="ignore"):
        keep = lnpdiff > np.log(u_acc)                                    # mh.py:157
    new_logp = logp.copy()
    new_logp[np.isinf(new_logp)] = 0.0                                    # move.py:513-532
    L[...] = logl * keep + L * (~keep)
    P[...] = new_logp * keep + P * (~keep)
    x[keep] = q[keep]
    return dict(q=q, logp=logp, logl=logl, lnpdiff=lnpdiff, keep=keep)


# --------------------------------------------------------------------------
# parallel tempering                                     moves/tempering.py
# --------------------------------------------------------------------------

def pt_sweep(x, L, P, betas, iperm, i1perm, u_swap):
    """Hot->cold swap cascade (moves/tempering.py:484-561, 351-482); in place.

    iperm, i1perm: int[T-1, W]; row j holds the draws for the pair (i, i-1)
    with i = T-1-j (the order they are drawn).  u_swap: float[T-1, W] likewise.
    Returns sel[T-1, W] (same row order) and swaps_accepted[T-1] indexed by i-1.
    """
    T, W, D = x.shape
    sel_all = np.zeros((T - 1, W), dtype=bool)
    swaps_accepted = np.empty(T - 1)
    for j, i in enumerate(range(T - 1, 0, -1)):
        dbeta = betas[i - 1] - betas[i]                               # :518-522
        ip, i1p = iperm[j], i1perm[j]
        with np.errstate(divide="ignore"):
            raccept = np.log(u_swap[j])                               # :535
        paccept = dbeta * (L[i, ip] - L[i - 1, i1p])                  # :538
        sel = paccept > raccept                                       # :541
        sel_all[j] = sel
        swaps_accepted[i - 1] = np.sum(sel)                           # :542
        a_, b_ = ip[sel], i1p[sel]
        for arr in (x, L, P):                                         # :376-480
            tmp = arr[i, a_].copy()
            arr[i, a_] = arr[i - 1, b_]
            arr[i - 1, b_] = tmp
    return sel_all, swaps_accepted


def adapt_ladder(betas, swaps_accepted, W, time, lag=10000, nu=100):
    """Ladder adaptation (moves/tempering.py:563-596).  Returns new betas."""
    ratios = swaps_accepted / np.full(len(betas) - 1, W)               # :587, :282
    b = betas.copy()
    decay = lag / (time + lag)                                        # :571
    kappa = decay / nu                                                # :572
    dSs = kappa * (ratios[:-1] - ratios[1:])                          # :575
    deltaTs = np.diff(1 / b[:-1])                                     # :578
    deltaTs *= np.exp(dSs)
    b[1:-1] = 1 / (np.cumsum(deltaTs) + 1 / b[0])                     # :580
    return betas + (b - betas)                                        # :583, :593


# --------------------------------------------------------------------------
# one full sampler iteration, driven by the two reference streams
# --------------------------------------------------------------------------

class OracleSampler:
    """Flat restatement of ``EnsembleSampler.sample``'s inner loop for one
    ``StretchMove`` (+ ``TemperatureControl``): ensemble.py:965-981,
    moves/red_blue.py:89-333, moves/tempering.py:598-649.
    """

    def __init__(self, x0, loglike, lo, hi, R, G, betas=None, a=2.0,
                 adaptive=True, permute=True, adaptation_lag=10000,
                 adaptation_time=100, stop_adaptation=-1, randomize_split=True,
                 live_dangerously=False, fill=-1e300, record=False, moves=None, period=None, nsplits=2):
        # moves: [("stretch" | GaussianProposal, weight), ...]; default the reference's single StretchMove
        self.nsplits = int(nsplits)                                   # RedBlueMove(nsplits=...), red_blue.py:41-47
        self.moves = [("stretch", 1.0)] if moves is None else list(moves)
        w = np.atleast_1d([m[1] for m in self.moves]).astype(float)
        self.weights = w / np.sum(w)                                  # ensemble.py:377-378
        self.move_accepted = None
        self.x = np.array(x0, dtype=np.float64, copy=True)
        self.T, self.W, self.D = self.x.shape
        self.loglike = loglike
        self.lo = np.asarray(lo, dtype=np.float64)
        self.hi = np.asarray(hi, dtype=np.float64)
        self.R, self.G = R, G
        self.tempered = betas is not None
        self.betas = None if betas is None else np.array(betas, dtype=np.float64, copy=True)
        self.a = a
        self.adaptive, self.permute = adaptive, permute
        self.lag, self.nu, self.stop_adaptation = adaptation_lag, adaptation_time, stop_adaptation
        self.randomize_split = randomize_split
        self.live_dangerously = live_dangerously
        self.fill = fill
        self.period = None if period is None else np.asarray(period, dtype=np.float64)   # periodic parameters (0: none)
        self.time = 0
        self.record = record
        self.trace = []
        T, W, D = self.T, self.W, self.D
        # initial log-prior / log-like (ensemble.py:898-912)
        self.P = box_log_prior(self.x.reshape(-1, D), self.lo, self.hi).reshape(T, W)
        self.L = compute_log_like(self.x, self.P, loglike, fill=fill)
        self.accepted = np.zeros((T, W))
        self.num_proposals = 0
        self.move_accepted = [np.zeros((T, W)) for _ in self.moves]
        self.move_num_proposals = [0 for _ in self.moves]
        self.swaps_accepted = np.zeros(max(T - 1, 0))
        self.swaps_accepted_total = np.zeros(max(T - 1, 0))

    # -- draws, in the reference's order ---------------------------------
    def draw_stretch(self, Ns, Nc):
        rint = self.R.randint(Nc, size=(self.T, Ns))                  # stretch.py:93-99
        u_zz = self.R.rand(self.T, Ns)                                # stretch.py:129-132
        return rint, u_zz

    def draw_pt(self):
        T, W = self.T, self.W
        iperm = np.empty((T - 1, W), dtype=np.int64)
        i1perm = np.empty((T - 1, W), dtype=np.int64)
        u = np.empty((T - 1, W))
        for j in range(T - 1):                                        # tempering.py:515-535
            if self.permute:
                iperm[j] = self.G.permutation(W)
                i1perm[j] = self.G.permutation(W)
            else:
                iperm[j] = np.arange(W)
                i1perm[j] = np.arange(W)
            u[j] = self.G.uniform(size=W)
        return iperm, i1perm, u

    def iteration(self):
        T, W, D = self.T, self.W, self.D
        rec = {}
        mi = int(self.R.choice(len(self.moves), p=self.weights))      # ensemble.py:971
        rec["move"] = mi
        if self.moves[mi][0] == "stretch":
            accepted = self._stretch_move(rec)
        else:
            accepted = self._mh_move(self.moves[mi][0], rec)
        self.accepted += accepted
        self.num_proposals += 1
        self.move_accepted[mi] += accepted
        self.move_num_proposals[mi] += 1
        if self.record:
            rec["L_stretch"], rec["P_stretch"] = self.L.copy(), self.P.copy()
        if self.tempered:
            iperm, i1perm, u_swap = self.draw_pt()
            sel, sw = pt_sweep(self.x, self.L, self.P, self.betas, iperm, i1perm, u_swap)
            self.swaps_accepted = sw
            self.swaps_accepted_total += sw
            if self.adaptive and T > 1:                               # tempering.py:632-633
                if self.stop_adaptation < 0 or self.time < self.stop_adaptation:
                    self.betas = adapt_ladder(self.betas, sw, W, self.time, self.lag, self.nu)
                self.time += 1
            if self.record:
                rec.update(iperm=iperm, i1perm=i1perm, u_swap=u_swap, sel=sel,
                           swaps_accepted=sw.copy(), betas_after=self.betas.copy())
        if self.record:
            rec.update(x=self.x.copy(), L=self.L.copy(), P=self.P.copy(), accepted=accepted)
            self.trace.append(rec)
        return accepted

    def _mh_move(self, prop, rec):
        """MHMove.propose (mh.py:56-193): proposal draws, then the accept uniforms, all from R."""
        T, W, D = self.T, self.W, self.D
        step = prop.draw_step(self.R, T * W, D).reshape(T, W, D)
        u_acc = self.R.rand(T, W)                                     # mh.py:157
        out = mh_step(self.x, self.L, self.P, self.betas, step, u_acc, self.lo, self.hi, self.loglike,
                      fill=self.fill, period=self.period)
        if self.record:
            rec.update(mh_step=step, mh_u_acc=u_acc, mh_q=out["q"], mh_logp=out["logp"], mh_logl=out["logl"],
                       mh_lnpdiff=out["lnpdiff"], mh_keep=out["keep"])
        return out["keep"]

    def _stretch_move(self, rec):
        T, W, D = self.T, self.W, self.D
        if W < 2 * D and not self.live_dangerously:                   # red_blue.py:108-114
            raise RuntimeError("It is unadvisable to use a red-blue move with fewer "
                               "walkers than twice the number of dimensions.")
        labels = split_labels(T, W, self.G, nsplits=self.nsplits, randomize=self.randomize_split)
        rec["labels"] = labels
        accepted = np.zeros((T, W), dtype=bool)
        tt = np.arange(T)[:, None]
        for split in range(self.nsplits):                             # red_blue.py:148
            Ns = int(np.sum(labels[0] == split))
            Nc = W - Ns
            rint, u_zz = self.draw_stretch(Ns, Nc)
            # u_acc is drawn after the likelihood in the reference (red_blue.py:294) but
            # nothing else touches R in between, so drawing it here is the same stream.
            u_acc = self.R.rand(T, Ns)
            out = stretch_split(self.x, self.L, self.P, self.betas, labels, split, rint,
                                u_zz, u_acc, self.a, self.lo, self.hi, self.loglike,
                                fill=self.fill, period=self.period, nsplits=self.nsplits)
            accepted[tt, out["S"]] = out["keep"]
            if self.record:
                rec[f"rint{split}"], rec[f"u_zz{split}"], rec[f"u_acc{split}"] = rint, u_zz, u_acc
                for k in ("q", "logp", "logl", "factors", "lnpdiff", "keep"):
                    rec[f"{k}{split}"] = out[k]
                rec[f"x_after{split}"] = self.x.copy()
        return accepted

    def run(self, n):
        for _ in range(n):
            self.iteration()
        return self
