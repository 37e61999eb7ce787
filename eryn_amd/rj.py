"""Reversible-jump leaf packing on the MI355X (BASELINE config 4).

Host-side mirror of the reference's multi-branch ``State`` (state.py:330-562: ``branches[name].coords[T, W, nleaves_max,
ndim]`` + ``branches[name].inds[T, W, nleaves_max]``) for the device path of ``include/hipensemble.h`` ``hens_rj_*``:
:class:`RJEngine` packs a variable-dimension ensemble into leaf-packing records, runs the in-model ``GaussianMove``
(mh.py:56-193, gaussian.py:68-270), the ``DistributionGenerateRJ`` birth / death move (distgenrj.py:35-222,
rj.py:145-388) and the PT sweeps on the device - teacher-forced with the reference's draws (parity) or with device-side
Philox draws (``step``) - and unpacks snapshots with the reference's NaN fill of unused leaves
(backends/backend.py:1049-1059).  The model is the reference tests' own template model (tests/test_eryn.py:38-92)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, f64, ptr
from .engine import HipEnsemble

KIND_PULSE, KIND_SINE = 0, 1
_KINDS = {"pulse": KIND_PULSE, "gauss": KIND_PULSE, "sine": KIND_SINE, KIND_PULSE: KIND_PULSE, KIND_SINE: KIND_SINE}


class TemplateBranch:
    """One model type: leaf kind ("pulse" | "sine"), a uniform box per leaf parameter, leaf budget."""

    def __init__(self, name, kind, box, nleaves_max, nleaves_min=0):
        self.name, self.kind = str(name), _KINDS[kind]
        self.lo = f64([b[0] for b in box])
        self.hi = f64([b[1] for b in box])
        if self.lo.shape != (3,):
            raise ValueError("a leaf has three parameters")
        self.ndim = 3
        self.nleaves_max, self.nleaves_min = int(nleaves_max), int(nleaves_min)
        acc = np.zeros(1)                      # prior.py:364-383: sequential ``prior_vals += temp`` from 0.0
        for d in range(3):
            acc += np.log(1 / (self.hi[d] - self.lo[d]))
        self.leaf_logp = float(acc[0])


class LeafBranch:
    """One model type WITHOUT a device likelihood (round 6): a uniform box per leaf parameter - 1 to 4 of them, the reference's
    ``ndims[name]`` (ensemble.py:325-329) - and a leaf budget.  For chains whose likelihood is the caller's Python function
    (CallableLikelihood, hens_rj_set_model_general)."""
    kind = 0

    def __init__(self, name, box, nleaves_max, nleaves_min=0):
        self.name = str(name)
        self.lo = f64([b[0] for b in box])
        self.hi = f64([b[1] for b in box])
        self.ndim = int(self.lo.shape[0])
        if not 1 <= self.ndim <= 4:
            raise NotImplementedError("a leaf has 1 to 4 parameters")
        self.nleaves_max, self.nleaves_min = int(nleaves_max), int(nleaves_min)
        acc = np.zeros(1)                      # prior.py:364-383: sequential ``prior_vals += temp`` from 0.0
        for d in range(self.ndim):
            acc += np.log(1 / (self.hi[d] - self.lo[d]))
        self.leaf_logp = float(acc[0])


class _TemplateLikelihood:
    kind = _lib.LIKE_TEMPLATE

    def __init__(self, ndim):
        self.ndim = ndim

    def _install(self, lib, ctx):
        pass


class CallableLikelihood:
    """A user ``log_like_fn`` for a state of several branches / leaves, called the way ``EnsembleSampler.compute_log_like`` calls
    it (ensemble.py:1219-1545): walkers with an infinite log-prior or without any active leaf are not evaluated; the active leaves
    of the others are packed per branch (``coords[inds]``: walker-major, slot order) with their group ids (utils/utility.py:8-40,
    renumbered 0..ngroups-1, ensemble.py:1306-1324); ``vectorize=True`` hands the function every group at once -
    ``fn([leaves_b0, leaves_b1, ...], [groups_b0, ...] if provide_groups, *args, **kwargs)`` - else it is called per group with
    ``[leaves_b or None for every branch]`` (:1420-1470).  Not evaluated -> ``fill_zero_leaves_val`` (:1486-1513); NaN raises.
    The proposal, prior, accept test and update around it run on the device (hens_rj_propose / hens_rj_accept)."""

    def __init__(self, fn, args=None, kwargs=None, vectorize=False, provide_groups=False, fill_zero_leaves_val=-1e300):
        self.fn, self.args, self.kwargs = fn, list(args or []), dict(kwargs or {})
        self.vectorize, self.provide_groups, self.fill = bool(vectorize), bool(provide_groups), float(fill_zero_leaves_val)
        self.ncalls = 0

    def __call__(self, x, inds, logp, names, only=None, has_reversible_jump=True):
        """x / inds: {name: [T, W, nl, ndim] / [T, W, nl]}, logp [T, W]; ``only`` [T, W] bool: the walkers a move touched (the
        others keep their log-likelihood: their entry here is never read)."""
        first = names[0]
        T, W = inds[first].shape[:2]
        for name in names:                                                  # ensemble.py:1258-1262
            xa = x[name][inds[name]]
            if np.any(np.isinf(xa)):
                raise ValueError("At least one parameter value was infinite")
            if np.any(np.isnan(xa)):
                raise ValueError("At least one parameter value was NaN")
        only = None if only is None else np.asarray(only, dtype=bool)
        if np.all(np.isinf(logp if only is None else logp[only])):          # :1272-1276
            import warnings
            warnings.warn("All points input for the Likelihood have a log prior of -inf.")
            return np.full_like(logp, -1e300)
        bad = np.isinf(logp) if only is None else (np.isinf(logp) | ~only)
        inds_copy = {k: np.array(inds[k], dtype=bool, copy=True) for k in names}
        for k in names:
            inds_copy[k][bad] = False
        gid = np.arange(T * W).reshape(T, W)
        groups = {k: np.repeat(gid[:, :, None], inds_copy[k].shape[2], axis=-1)[inds_copy[k]] for k in names}
        unique_groups = np.unique(np.concatenate([groups[k] for k in names]))
        groups_map = np.arange(len(unique_groups))
        ll_groups = []
        for k in names:
            tu, inverse = np.unique(groups[k], return_inverse=True)
            ll_groups.append(groups_map[np.isin(unique_groups, tu)][inverse])
        params_in = [x[k][inds_copy[k]] for k in names]
        ll = np.full(T * W, -1e300)
        if len(unique_groups):
            if self.vectorize:
                a = [params_in[0] if len(params_in) == 1 else params_in]
                if self.provide_groups:
                    a.append(ll_groups[0] if len(ll_groups) == 1 else ll_groups)
                results = self.fn(*a, *self.args, **self.kwargs)
                self.ncalls += 1
            else:
                results = []
                for g in groups_map:
                    arg = [None] * len(names)
                    for bi in range(len(names)):
                        keep = np.where(ll_groups[bi] == g)[0]
                        if keep.shape[0] > 0:
                            p = params_in[bi][keep]
                            if not has_reversible_jump and p.shape[0] == 1:
                                p = p[0]
                            arg[bi] = p
                    results.append(self.fn(arg[0] if len(names) == 1 else arg, *self.args, **self.kwargs))
                    self.ncalls += 1
            results = np.asarray(results)
            if results.ndim == 2 and results.shape[1] == 1:
                results = np.squeeze(results, axis=1)
            if results.ndim != 1:
                raise NotImplementedError("blobs are outside the device hot path: the likelihood must return one value per group")
            ll[unique_groups] = results
        ll[np.delete(np.arange(T * W), unique_groups)] = self.fill
        if np.any(np.isnan(ll)):
            raise ValueError("The likelihood function is returning Nan.")
        return ll.reshape(T, W)


class RJEngine:
    def __init__(self, ntemps, nwalkers, branches, t, y, sigma, seed=0, device_id=0, adaptive=True,
                 adaptation_lag=10000, adaptation_time=100, stop_adaptation=-1, fill_value=-1e300, a=2.0, live_dangerously=False):
        self.branches = list(branches)
        if not 1 <= len(self.branches) <= 4:
            raise NotImplementedError("1 to 4 branches")
        self.ncoord = sum(b.nleaves_max * b.ndim for b in self.branches)
        self.ndmax = max(b.ndim for b in self.branches)          # (the stride of the birth arrays: 3 for the template models)
        # a model without a device likelihood: any LeafBranch, or no data (hens_rj_set_model_general; host_like steps it)
        self.general = t is None or any(not isinstance(b, TemplateBranch) for b in self.branches)
        rw = self.ncoord + len(self.branches)
        self.RW = rw + (rw & 1)                                  # even record width: 16-byte row alignment
        if self.RW > 128:
            raise NotImplementedError(f"{self.ncoord} leaf coordinates + {len(self.branches)} masks: a record holds 128 doubles")
        self.off = np.cumsum([0] + [b.nleaves_max * b.ndim for b in self.branches])[:-1]
        self.T, self.W = int(ntemps), int(nwalkers)
        # the engine's prior box is unused on records; HipEnsemble wants one
        self.eng = HipEnsemble(self.T, self.W, self.RW, _TemplateLikelihood(self.RW), -1.0, 1.0, tempered=True,
                               adaptive=adaptive, adaptation_lag=adaptation_lag, adaptation_time=adaptation_time,
                               stop_adaptation=stop_adaptation, live_dangerously=live_dangerously, fill_value=fill_value, seed=seed,
                               device_id=device_id, a=a)
        self.lib, self.ctx = self.eng.lib, self.eng.ctx
        self.schedule = "separate_branches"
        self.host_like = None        # a CallableLikelihood: every parity move = propose on the device, evaluate here, accept on the device
        nb = len(self.branches)
        kinds = np.array([b.kind for b in self.branches], dtype=np.int32)
        nlmax = np.array([b.nleaves_max for b in self.branches], dtype=np.int32)
        nlmin = np.array([b.nleaves_min for b in self.branches], dtype=np.int32)
        lp = f64([b.leaf_logp for b in self.branches])
        if self.general:
            nds = np.array([b.ndim for b in self.branches], dtype=np.int32)
            lo, hi = f64(np.concatenate([b.lo for b in self.branches])), f64(np.concatenate([b.hi for b in self.branches]))
            check(self.lib.hens_rj_set_model_general(self.ctx, nb, ptr(nds), ptr(nlmax), ptr(nlmin), ptr(lo), ptr(hi), ptr(lp)), self.ctx)
            return
        lo = f64(np.stack([b.lo for b in self.branches]))
        hi = f64(np.stack([b.hi for b in self.branches]))
        t, y = f64(t), f64(y)
        if t.shape != y.shape or t.ndim != 1:
            raise ValueError("t and y must be 1-D arrays of the same length")
        check(self.lib.hens_rj_set_model(self.ctx, nb, ptr(kinds), ptr(nlmax), ptr(nlmin), ptr(lo), ptr(hi), ptr(lp),
                                         int(t.shape[0]), ptr(t), ptr(y), float(sigma)), self.ctx)

    def close(self):
        self.eng.close()

    # -- records <-> branches --------------------------------------------------------------------------------
    def pack(self, x, inds):
        """{name: coords[T, W, nl, ndim]}, {name: inds[T, W, nl]} -> records[T, W, RW]."""
        rec = np.zeros((self.T, self.W, self.RW))
        for bi, b in enumerate(self.branches):
            c = np.asarray(x[b.name], dtype=np.float64)
            if c.shape != (self.T, self.W, b.nleaves_max, b.ndim):
                raise ValueError(f"coords of branch {b.name} must have shape {(self.T, self.W, b.nleaves_max, b.ndim)}")
            c = np.where(np.isnan(c), 0.0, c)                    # a stored chain marks unused leaves with NaN
            rec[:, :, self.off[bi]:self.off[bi] + b.nleaves_max * b.ndim] = c.reshape(self.T, self.W, -1)
            m = np.asarray(inds[b.name], dtype=bool)
            rec[:, :, self.ncoord + bi] = (m * (1 << np.arange(b.nleaves_max))).sum(axis=-1)
        return rec

    def unpack(self, rec, nan_fill=False):
        x, inds = {}, {}
        for bi, b in enumerate(self.branches):
            x[b.name] = rec[:, :, self.off[bi]:self.off[bi] + b.nleaves_max * b.ndim].reshape(self.T, self.W, b.nleaves_max, b.ndim).copy()
            m = rec[:, :, self.ncoord + bi].astype(np.int64)
            inds[b.name] = ((m[:, :, None] >> np.arange(b.nleaves_max)) & 1).astype(bool)
            if nan_fill:                                         # backend.py:1049-1059
                x[b.name][~inds[b.name]] = np.nan
        return x, inds

    def steps_to_records(self, steps):
        """{name: step[T, W, nl, ndim]} (zero on unused slots) -> [T, W, ncoord] in record layout."""
        out = np.zeros((self.T, self.W, self.ncoord))
        for bi, b in enumerate(self.branches):
            out[:, :, self.off[bi]:self.off[bi] + b.nleaves_max * b.ndim] = np.asarray(steps[b.name]).reshape(self.T, self.W, -1)
        return out

    # -- state ---------------------------------------------------------------------------------------------------
    def upload(self, x, inds, logl=None, logp=None, betas=None):
        self.eng.upload(self.pack(x, inds), logl, logp, betas)

    def download(self, nan_fill=False):
        rec, L, P, betas = self.eng.download()
        x, inds = self.unpack(rec, nan_fill=nan_fill)
        return x, inds, L, P, betas

    def eval_state(self):
        self.eng.eval_state()                                    # (log-prior on the device; log-like: the template model's)
        if self.host_like is not None:                           # ... or the host callable's
            rec, _, P, betas = self.eng.download()
            x, inds = self.unpack(rec)
            L = self.host_like(x, inds, P, [b.name for b in self.branches])
            self.eng.upload(rec, L, P, betas)

    def set_adapt_time(self, t):
        self.eng.set_adapt_time(t)

    # -- moves with a host-callable likelihood: hens_rj_propose -> the user's function -> hens_rj_accept -------------
    def _host_move(self, move, **d):
        """One teacher-forced move whose likelihood is ``self.host_like``: the device proposes and computes the log-prior, the
        host packs the active leaves and calls the user's function (CallableLikelihood), the device tests and updates."""
        self.eng.state_epoch += 1
        keepalive = {k: v for k, v in d.items() if isinstance(v, np.ndarray)}
        dr = _lib.HensRjDraws(**{k: (v.ctypes.data if isinstance(v, np.ndarray) else v) for k, v in d.items()})
        q = np.empty((self.T, self.W, self.RW))
        logp = np.empty((self.T, self.W))
        moved = np.empty((self.T, self.W), dtype=np.uint8)
        check(self.lib.hens_rj_propose(self.ctx, int(move), C.byref(dr), ptr(q), ptr(logp), ptr(moved)), self.ctx)
        del keepalive
        x, inds = self.unpack(q)
        try:
            logl = f64(self.host_like(x, inds, logp, [b.name for b in self.branches], only=moved.astype(bool)))
        except Exception:
            # the move must not stay half done: reject everything (log-like -inf fails every accept test), then re-raise
            check(self.lib.hens_rj_accept(self.ctx, ptr(np.full((self.T, self.W), -np.inf)), None), self.ctx)
            raise
        keep = np.empty((self.T, self.W), dtype=np.uint8)
        check(self.lib.hens_rj_accept(self.ctx, ptr(logl), ptr(keep)), self.ctx)
        return keep.astype(bool)

    # -- parity-mode moves -----------------------------------------------------------------------------------------
    def mh_step(self, steps, u_acc):
        st = f64(self.steps_to_records(steps))
        u = f64(u_acc, (self.T, self.W))
        if self.host_like is not None:
            return self._host_move(_lib.RJ_MOVE_MH, step=st, u_acc=u)
        keep = np.empty((self.T, self.W), dtype=np.uint8)
        check(self.lib.hens_rj_mh_step(self.ctx, ptr(st), ptr(u), ptr(keep)), self.ctx)
        return keep.astype(bool)

    def bd_step(self, branch, change, leaf, birth, u_acc):
        """change[T, W] in {-1, 0, +1}, leaf[T, W] slot, birth[T, W, ndim of the branch] (rows of walkers that give birth), u_acc[T, W]."""
        ch = np.ascontiguousarray(change, dtype=np.int8)
        lf = np.ascontiguousarray(np.where(np.asarray(change) == 0, 0, leaf), dtype=np.int32)
        bt = self._birth_rows(birth, self.branches[int(branch)].ndim)
        u = f64(u_acc, (self.T, self.W))
        if self.host_like is not None:
            return self._host_move(_lib.RJ_MOVE_BD, branch=int(branch), change=ch, leaf=lf, birth=bt, u_acc=u)
        keep = np.empty((self.T, self.W), dtype=np.uint8)
        check(self.lib.hens_rj_bd_step(self.ctx, int(branch), ptr(ch), ptr(lf), ptr(bt), ptr(u), ptr(keep)), self.ctx)
        return keep.astype(bool)

    def bd_all_step(self, change, leaf, birth, u_acc):
        """"together": one proposal over every branch - change / leaf [nbranches, T, W], birth [nbranches][T, W, ndim of the branch]
        (one array when every branch has the same width), u_acc [T, W]."""
        nb = len(self.branches)
        ch = np.ascontiguousarray(change, dtype=np.int8)
        lf = np.ascontiguousarray(np.where(np.asarray(change) == 0, 0, leaf), dtype=np.int32)
        bt = np.ascontiguousarray(np.stack([self._birth_rows(birth[bi], b.ndim) for bi, b in enumerate(self.branches)]))
        u = f64(u_acc, (self.T, self.W))
        if ch.shape != (nb, self.T, self.W) or lf.shape != ch.shape:
            raise ValueError("change / leaf must have shape (nbranches, ntemps, nwalkers)")
        if self.host_like is not None:
            return self._host_move(_lib.RJ_MOVE_BD_ALL, change=ch, leaf=lf, birth=bt, u_acc=u)
        keep = np.empty((self.T, self.W), dtype=np.uint8)
        check(self.lib.hens_rj_bd_all_step(self.ctx, ptr(ch), ptr(lf), ptr(bt), ptr(u), ptr(keep)), self.ctx)
        return keep.astype(bool)

    def _birth_rows(self, birth, nd):
        """[T, W, nd] -> [T, W, ndmax] (the library's stride: the model's widest branch)."""
        b = f64(birth, (self.T, self.W, nd))
        if nd == self.ndmax:
            return b
        out = np.zeros((self.T, self.W, self.ndmax))
        out[:, :, :nd] = b
        return out

    def stretch_split(self, split, labels, rint, u_zz, u_acc):
        """One half of the red / blue StretchMove over every branch and leaf slot (stretch.py:160-231, red_blue.py:148-323):
        labels[T, W] in {0, 1}, rint[nbranches, T, Ns] (one complement draw per branch), u_zz / u_acc[T, Ns]; returns the accept
        mask [T, Ns] of the moving walkers in ascending order."""
        lab = np.ascontiguousarray(labels, dtype=np.uint8)
        ri = np.ascontiguousarray(rint, dtype=np.int64)
        Ns = ri.shape[-1]
        if lab.shape != (self.T, self.W) or ri.shape != (len(self.branches), self.T, Ns):
            raise ValueError("labels must have shape (ntemps, nwalkers), rint (nbranches, ntemps, Ns)")
        uz, ua = f64(u_zz, (self.T, Ns)), f64(u_acc, (self.T, Ns))
        if self.host_like is not None:
            kw = self._host_move(_lib.RJ_MOVE_STRETCH, split=int(split), labels=lab, rint=ri, u_zz=uz, u_acc=ua)     # by walker
            return np.stack([kw[t][lab[t] == split] for t in range(self.T)])
        keep = np.empty((self.T, Ns), dtype=np.uint8)
        check(self.lib.hens_rj_stretch_split(self.ctx, int(split), ptr(lab), ptr(ri), ptr(uz), ptr(ua), ptr(keep)), self.ctx)
        return keep.astype(bool)

    def pt_sweep(self, iperm, i1perm, u_swap, adapt=True):
        return self.eng.pt_sweep(iperm, i1perm, u_swap, adapt=adapt)

    # -- production -------------------------------------------------------------------------------------------------
    def set_mh_scale(self, scale):
        """Standard deviations of the in-model Gaussian step per branch and leaf parameter: [nbranches, 3] (template models)."""
        s = f64(scale, (len(self.branches), 3))
        check(self.lib.hens_rj_set_mh_scale(self.ctx, ptr(s)), self.ctx)

    def step(self, n_iters):
        check(self.lib.hens_rj_step(self.ctx, int(n_iters)), self.ctx)

    def set_schedule(self, rj_moves):
        """The sampler's ``rj_moves`` string for ``step`` (ensemble.py:434-480): "separate_branches" | "iterate_branches"."""
        code = {"separate_branches": 0, "iterate_branches": 1, "together": 2}.get(rj_moves)
        if code is None:
            raise ValueError("rj_moves must be 'together', 'iterate_branches', or 'separate_branches'")
        check(self.lib.hens_rj_set_schedule(self.ctx, code), self.ctx)
        self.schedule = rj_moves

    def debug_draws(self, it):
        """Everything ``step`` draws in iteration ``it`` (include/hipensemble.h: hens_rj_debug_draws).  The birth / death
        arrays carry a leading branch axis: one entry ("separate_branches": the chosen branch) or one per branch in order."""
        T, W = self.T, self.W
        ns = len(self.branches)
        out = dict(step=np.empty((T, W, self.ncoord)), u_mh=np.empty((T, W)), coin=np.empty((ns, T, W), dtype=np.int8),
                   sel=np.empty((ns, T, W), dtype=np.uint32), birth=np.empty((ns, T, W, 3)), u_bd=np.empty((ns, T, W)),
                   slot_mh=np.empty((T, W), dtype=np.int32), uswap_mh=np.empty((max(T - 1, 1), W)),
                   slot_bd=np.empty((T, W), dtype=np.int32), uswap_bd=np.empty((max(T - 1, 1), W)))
        br = C.c_int32(0)
        check(self.lib.hens_rj_debug_draws(self.ctx, int(it), ptr(out["step"]), ptr(out["u_mh"]), C.byref(br), ptr(out["coin"]),
                                           ptr(out["sel"]), ptr(out["birth"]), ptr(out["u_bd"]), ptr(out["slot_mh"]),
                                           ptr(out["uswap_mh"]), ptr(out["slot_bd"]), ptr(out["uswap_bd"])), self.ctx)
        out["branch"] = int(br.value)
        if out["branch"] >= 0:                          # one move on the chosen branch
            for k in ("coin", "sel", "birth", "u_bd"):
                out[k] = out[k][:1]
        return out

    def iteration(self):
        return self.eng.iteration()

    def synchronize(self):
        self.eng.synchronize()

    def counters(self):
        c = self.eng.counters()
        bd = np.empty((self.T, self.W))
        n_mh, n_bd = C.c_int64(0), C.c_int64(0)
        check(self.lib.hens_rj_get_counters(self.ctx, ptr(bd), C.byref(n_mh), C.byref(n_bd)), self.ctx)
        c.update(accepted_mh=c["accepted"], accepted_bd=bd, num_mh=int(n_mh.value), num_bd=int(n_bd.value))
        return c


# ---------------------------------------------------------------------------------------------------------------------
# Sampler-level mirror: the reference's multi-branch ``EnsembleSampler`` contract for this path
# ---------------------------------------------------------------------------------------------------------------------
class TemplateLikelihood:
    """Stands where the reference takes ``log_like_fn`` + ``args=[t, y, sigma]`` (tests/test_eryn.py:79-92, 467-470):
    the template model lives inside the kernel.  ``kinds``: ``{branch_name: "pulse" | "sine"}``."""

    def __init__(self, kinds, t, y, sigma):
        self.kinds = {k: _KINDS[v] for k, v in kinds.items()}
        self.t, self.y, self.sigma = (None if t is None else f64(t)), (None if y is None else f64(y)), float(sigma)


class GaussianLeafMove:
    """``GaussianMove(cov_all)`` of the reference on leaf-packing records (gaussian.py:9-66): ``cov_all[name]`` is the
    3 x 3 proposal covariance of a leaf of that branch (the reference's ``_proposal``: ``multivariate_normal``)."""

    def __init__(self, cov_all):
        self.cov = {k: np.atleast_2d(np.asarray(v, dtype=np.float64)) for k, v in cov_all.items()}
        for k, c in self.cov.items():
            if c.ndim != 2 or c.shape[0] != c.shape[1] or not 1 <= c.shape[0] <= 4:
                raise NotImplementedError("a leaf's proposal covariance must be an ndim x ndim matrix (ndim = 1 .. 4)")
        self.accepted, self.num_proposals = None, 0


class StretchLeafMove:
    """``StretchMove(a=2.0)`` of the reference as the in-model move on leaf-packing records (stretch.py:14-231 over
    red_blue.py:103-330): every branch and every leaf slot of a walker moves - one complement walker per branch, one stretch
    factor per walker.  (The reference advises against it beside reversible jump, ensemble.py:509-514, and runs it.)"""

    def __init__(self, a=2.0, live_dangerously=False):
        self.a = float(a)
        self.live_dangerously = bool(live_dangerously)         # red_blue.py:41-47,108: fewer walkers than 2 x (all leaf slots' coordinates)
        self.accepted, self.num_proposals = None, 0


class RJEnsembleSampler:
    """``EnsembleSampler(nwalkers, ndims, log_like_fn, priors, tempering_kwargs=..., branch_names=..., nleaves_max=...,
    nleaves_min=..., moves=GaussianMove(cov), rj_moves="separate_branches")`` (ensemble.py:211-681) for the template model,
    stepping on the MI355X.  One iteration = one in-model move + one RJ move of a uniformly chosen branch
    (ensemble.py:963-1024 with the default ``num_repeats_in_model = num_repeats_rj = 1``).

    rng="numpy":  the reference's streams (the sampler-owned RandomState cloned from the global ``np.random`` at
                  construction + the global stream) are drawn on the host IN THE REFERENCE'S ORDER and handed to the device:
                  same seeds => the reference's chain (tests/test_hip_rj.py against the rj* fixtures).
    rng="philox": device-side draws, ``thin_by`` iterations per host call (``hens_rj_step``).  The device keeps every walker's
                  model resident and updates it by +- one leaf per accepted birth / death; the models and log-likelihoods are
                  re-evaluated from the coordinates every 64 iterations and whenever the state is downloaded (the evaluation
                  runs in front of the copy, so a stored ``State.log_like`` is exactly what the device continues with and a chain
                  resumed from it is the uninterrupted chain bit for bit).  Consequence: WHERE the re-evaluations fall depends
                  on ``store`` / ``thin_by``, so two runs that differ only in those agree to rounding (log-likelihoods to ~1e-13
                  relative), not bit for bit - as two reference runs with a different likelihood summation order would."""

    def __init__(self, nwalkers, ndims, log_like_fn, priors, tempering_kwargs=None, nbranches=None, branch_names=None,
                 nleaves_max=None, nleaves_min=None, moves=None, rj_moves="separate_branches", rng="numpy", seed=None,
                 device_id=0, args=None, kwargs=None, vectorize=False, provide_groups=False, fill_zero_leaves_val=-1e300, **unused):
        from .moves.tempering import TemperatureControl
        # log_like_fn: a TemplateLikelihood (the model lives in the kernel) or - round 6 - any Python function of the packed active
        # leaves with the reference's own ``args`` / ``kwargs`` / ``vectorize`` / ``provide_groups`` (ensemble.py:211-330): the
        # device proposes, computes the prior, tests and updates, the host evaluates (CallableLikelihood)
        self.host_like = None
        if callable(log_like_fn) and not isinstance(log_like_fn, (TemplateLikelihood, CallableLikelihood)):
            log_like_fn = CallableLikelihood(log_like_fn, args, kwargs, vectorize, provide_groups, fill_zero_leaves_val)
        if isinstance(log_like_fn, CallableLikelihood):
            if rng != "numpy":
                raise NotImplementedError("a host-callable likelihood steps with rng='numpy' (rng='philox' needs the likelihood on the device)")
            self.host_like = log_like_fn
            names_ = list(branch_names if branch_names is not None else ndims.keys())
            log_like_fn = TemplateLikelihood({k: "pulse" for k in names_}, None, None, 1.0)     # (never evaluated: no device likelihood)
        if not isinstance(log_like_fn, TemplateLikelihood):
            raise NotImplementedError("log_like_fn: an eryn_amd.rj.TemplateLikelihood, a CallableLikelihood or a Python function")
        if rj_moves not in ("separate_branches", "iterate_branches", "together", None, False):
            raise ValueError("When providing a str for rj_moves, must be 'together', 'iterate_branches', or "
                             f"'separate_branches'. Input is {rj_moves}")                # ensemble.py:473-476
        self.rj_schedule = rj_moves or None                  # None: no reversible jump - the leaf masks never change
        if not isinstance(moves, (GaussianLeafMove, StretchLeafMove)):
            raise NotImplementedError("the in-model move must be an eryn_amd.rj.GaussianLeafMove or StretchLeafMove")
        if rng not in ("numpy", "philox"):
            raise ValueError("rng must be 'numpy' or 'philox'")
        if isinstance(moves, StretchLeafMove) and rng != "numpy":
            raise NotImplementedError("the stretch move on leaf-packing records steps with rng='numpy' (the reference's draws)")
        if self.rj_schedule is None and rng != "numpy":
            raise NotImplementedError("rng='philox' steps the in-model move and the birth / death move together (hens_rj_step)")
        self.branch_names = list(branch_names if branch_names is not None else ndims.keys())
        if nbranches is not None and nbranches != len(self.branch_names):
            raise ValueError("nbranches does not match branch_names")
        nleaves_min = nleaves_min or {k: 0 for k in self.branch_names}            # ensemble.py:388-389
        self.nwalkers, self.ndims = int(nwalkers), dict(ndims)
        self.nleaves_max, self.nleaves_min = dict(nleaves_max), dict(nleaves_min)
        self.branches = []
        for k in self.branch_names:
            nd = int(self.ndims[k])
            if self.host_like is None and nd != 3:
                raise NotImplementedError("a leaf of the template model has three parameters")
            if isinstance(moves, GaussianLeafMove) and moves.cov[k].shape != (nd, nd):
                raise ValueError(f"branch {k}: the proposal covariance must be {nd} x {nd}")
            pr = priors[k]
            box = pr.box_bounds() if hasattr(pr, "box_bounds") else \
                ([pr[i].min_val for i in range(nd)], [pr[i].max_val for i in range(nd)])
            if len(box[0]) != nd:
                raise ValueError(f"branch {k}: {len(box[0])} prior boxes for ndims = {nd}")
            if self.host_like is not None:     # (no device likelihood: 1 .. 4 parameters per leaf, hens_rj_set_model_general)
                self.branches.append(LeafBranch(k, list(zip(box[0], box[1])), self.nleaves_max[k], self.nleaves_min[k]))
            else:
                self.branches.append(TemplateBranch(k, log_like_fn.kinds[k], list(zip(box[0], box[1])), self.nleaves_max[k],
                                                    self.nleaves_min[k]))
        total_ndim = sum(self.nleaves_max[k] * self.ndims[k] for k in self.branch_names)     # ensemble.py:325-329
        tk = dict(tempering_kwargs or {})
        if not tk:
            raise NotImplementedError("the device RJ path is tempered (pass tempering_kwargs=dict(ntemps=...))")
        self.temperature_control = tc = TemperatureControl(total_ndim, self.nwalkers, **tk)
        self.ntemps = tc.ntemps
        self.moves, self.rng = [moves], rng
        if seed is None:
            seed = int(np.random.randint(0, 2**31 - 1)) if rng == "philox" else 0
        self.engine = RJEngine(self.ntemps, self.nwalkers, self.branches, log_like_fn.t, log_like_fn.y, log_like_fn.sigma,
                               seed=seed, device_id=device_id, adaptive=tc.adaptive, adaptation_lag=tc.adaptation_lag,
                               adaptation_time=tc.adaptation_time, stop_adaptation=tc.stop_adaptation,
                               **({"a": moves.a, "live_dangerously": moves.live_dangerously} if isinstance(moves, StretchLeafMove) else {}))
        self.engine.host_like = self.host_like
        if self.rj_schedule is not None:
            self.engine.set_schedule(rj_moves)
        if rng == "philox":
            # the device draws axis-aligned steps (hens_rj_set_mh_scale: three standard deviations per branch): a covariance
            # with off-diagonal terms is another proposal - refused rather than silently reduced to its diagonal
            for k in self.branch_names:
                if np.any(moves.cov[k] != np.diag(np.diag(moves.cov[k]))):
                    raise NotImplementedError("rng='philox' takes diagonal leaf covariances (rng='numpy' runs the reference's "
                                              "multivariate_normal draws with any covariance)")
            self.engine.set_mh_scale(np.stack([np.sqrt(np.diag(moves.cov[k])) for k in self.branch_names]))
        moves.accepted = np.zeros((self.ntemps, self.nwalkers))
        nmoves = len(self.branch_names) if rj_moves == "separate_branches" else (1 if self.rj_schedule else 0)    # (rj move objects, ensemble.py:414-471)
        self.rj_accepted = [np.zeros((self.ntemps, self.nwalkers)) for _ in range(nmoves)]
        self.rj_num_proposals = [0 for _ in range(nmoves)]
        # rng="philox": the device counts the birth / death move over all branches together
        self.rj_accepted_all, self.rj_num_proposals_all = np.zeros((self.ntemps, self.nwalkers)), 0
        self._random = np.random.mtrand.RandomState()
        self._random.set_state(np.random.get_state())          # R := snapshot of the global stream (ensemble.py:604,651-652)
        self.iteration, self.chain = 0, []
        self._previous_state = None

    # -- the reference's evaluation entry points, with inds (ensemble.py:1127-1217, 1219-1545) -----------------------
    def _eval(self, coords, inds):
        self.engine.upload(coords, inds, betas=self.temperature_control.betas)
        self.engine.eval_state()                 # (a host-callable likelihood is called from there)
        _, _, L, P, _ = self.engine.download()
        return L, P

    def compute_log_prior(self, coords, inds=None, **kw):
        return self._eval(coords, inds)[1]

    def compute_log_like(self, coords, inds=None, logp=None, **kw):
        return self._eval(coords, inds)[0], None

    # -- one iteration with the reference's draws ------------------------------------------------------------------------
    def _iteration_numpy(self):
        eng, tc, R, T, W = self.engine, self.temperature_control, self._random, self.ntemps, self.nwalkers
        mv = self.moves[0]
        R.choice(1, p=np.ones(1))                                               # move choice (ensemble.py:971)
        if isinstance(mv, StretchLeafMove):
            # red / blue stretch over every branch and leaf slot: the split's shuffles from the GLOBAL stream (red_blue.py:119-124),
            # per half and branch one randint of R, behind the first branch's the walkers' zz uniforms (stretch.py:205, 128-132),
            # then the accept uniforms (red_blue.py:294)
            labels = np.tile(np.arange(W), (T, 1)) % 2
            for row in labels:
                np.random.shuffle(row)
            acc = np.zeros((T, W), dtype=bool)
            for split in range(2):
                Ns = int((labels[0] == split).sum())
                rint, u_zz = [], None
                for bi in range(len(self.branches)):
                    rint.append(R.randint(W - Ns, size=(T, Ns)))
                    if bi == 0:
                        u_zz = R.rand(T, Ns)
                keep = eng.stretch_split(split, labels, np.stack(rint), u_zz, R.rand(T, Ns))
                for t in range(T):
                    acc[t, np.flatnonzero(labels[t] == split)] = keep[t]
        else:
            x, inds, _, _, _ = eng.download()
            # per branch one multivariate_normal over the packed leaves (gaussian.py:96-104, 265-268), the accept uniforms (mh.py:157)
            steps = {}
            for b in self.branches:
                n = int(inds[b.name].sum())
                s = np.zeros(x[b.name].shape)
                s[inds[b.name]] = 1.0 * R.multivariate_normal(np.zeros(b.ndim), mv.cov[b.name], size=n)
                steps[b.name] = s
            acc = eng.mh_step(steps, R.rand(T, W))
        mv.accepted += acc
        mv.num_proposals += 1
        iperm, i1perm, u = tc.draw_swap_randoms()                               # mh.py:190-191
        _, swaps = eng.pt_sweep(iperm, i1perm, u, adapt=bool(tc.adaptive))
        tc.swaps_accepted = swaps
        if tc.adaptive:
            tc.time += 1
        if self.rj_schedule is None:
            return acc, None
        # reversible jump (ensemble.py:988-990; distgenrj.py:35-222): on one branch chosen from R, or - "iterate_branches" - one
        # move (the choice among ONE move still draws) that takes the branches in turn, its accept mask the last branch's
        nb = len(self.branches)
        if self.rj_schedule == "together":
            R.choice(1, p=np.ones(1))
            racc = self._bd_numpy_all()
            self.rj_accepted[0] += racc
            self.rj_num_proposals[0] += 1
        elif self.rj_schedule == "iterate_branches":
            R.choice(1, p=np.ones(1))
            for bi in range(nb):
                racc = self._bd_numpy(bi)
            self.rj_accepted[0] += racc
            self.rj_num_proposals[0] += 1
        else:
            bi = int(R.choice(nb, p=np.full(nb, 1.0 / nb)))
            racc = self._bd_numpy(bi)
            self.rj_accepted[bi] += racc
            self.rj_num_proposals[bi] += 1
        iperm, i1perm, u = tc.draw_swap_randoms()
        eng.pt_sweep(iperm, i1perm, u, adapt=False)                             # rj.py:381-382
        return acc, racc

    def _bd_numpy_all(self):
        """"together": every branch in one proposal - all branches' coins and leaf choices from R first, then the births from
        the global stream branch by branch (distgenrj.py:166-220)."""
        eng, tc, R, T, W = self.engine, self.temperature_control, self._random, self.ntemps, self.nwalkers
        _, inds, _, _, betas = eng.download()
        tc.betas = betas
        nb = len(self.branches)
        change, leaf = np.zeros((nb, T, W), dtype=np.int64), np.zeros((nb, T, W), dtype=np.int64)
        birth = [np.zeros((T, W, b.ndim)) for b in self.branches]
        for bi, b in enumerate(self.branches):
            if b.nleaves_min != b.nleaves_max:
                change[bi], leaf[bi] = self._draw_change_leaf(b, inds[b.name])
        for bi, b in enumerate(self.branches):
            if b.nleaves_min != b.nleaves_max:
                birth[bi][change[bi] == +1] = self._draw_births(b, int((change[bi] == +1).sum()))
        return eng.bd_all_step(change, leaf, birth, R.rand(T, W))               # rj.py:332

    def _draw_change_leaf(self, b, ib):
        R, T, W = self._random, self.ntemps, self.nwalkers
        nleaves = ib.sum(axis=-1)
        leaf = np.zeros((T, W), dtype=np.int64)
        change = R.choice([-1, +1], size=nleaves.shape)                         # distgenrj.py:63-66
        change = (change * ((nleaves != b.nleaves_min) & (nleaves != b.nleaves_max))
                  + (+1) * (nleaves == b.nleaves_min) + (-1) * (nleaves == b.nleaves_max))       # :69-73
        for t in range(T):                                                      # one draw per walker, in order (:85-121)
            for w in range(W):
                if change[t, w] == +1:
                    leaf[t, w] = R.choice(np.where(~ib[t, w])[0])
                elif change[t, w] == -1:
                    leaf[t, w] = R.choice(np.where(ib[t, w])[0])
        return change, leaf

    @staticmethod
    def _draw_births(b, nbirth):
        draws = np.zeros((nbirth, b.ndim))
        for d in range(b.ndim):                                                 # ProbDistContainer.rvs: GLOBAL stream, per parameter
            draws[:, d] = np.random.rand(nbirth) * (b.hi[d] - b.lo[d]) + b.lo[d]          # prior.py:60-66, 432-497
        return draws

    def _bd_numpy(self, bi):
        """Birth / death on branch ``bi`` with the reference's draws (distgenrj.py:35-222, rj.py:169-352)."""
        eng, tc, R, T, W = self.engine, self.temperature_control, self._random, self.ntemps, self.nwalkers
        _, inds, _, _, betas = eng.download()
        tc.betas = betas
        b = self.branches[bi]
        change, leaf, birth = np.zeros((T, W), dtype=np.int64), np.zeros((T, W), dtype=np.int64), np.zeros((T, W, b.ndim))
        if b.nleaves_min != b.nleaves_max:
            change, leaf = self._draw_change_leaf(b, inds[b.name])
            birth[change == +1] = self._draw_births(b, int((change == +1).sum()))
        return eng.bd_step(bi, change, leaf, birth, R.rand(T, W))               # rj.py:332

    def _state(self, nan_fill=False):
        from .state import State
        x, inds, L, P, betas = self.engine.download(nan_fill=nan_fill)
        return State(x, inds=inds, log_like=L, log_prior=P, betas=betas)

    def run_mcmc(self, initial_state, nsteps, burn=None, thin_by=1, store=True, **unused):
        """ensemble.py:1047-1125; returns the last State.  Stored steps keep the reference's NaN fill of unused leaves
        (backends/backend.py:1049-1059) in ``self.chain`` (a list of States)."""
        from .state import State
        tc = self.temperature_control
        if initial_state is None:
            if self._previous_state is None:
                raise ValueError("Cannot have `initial_state=None` if run_mcmc has never been called.")
            initial_state = self._previous_state
        st = State(initial_state, copy=True)
        coords, inds = st.branches_coords, st.branches_inds
        for b in self.branches:
            if coords[b.name].shape != (self.ntemps, self.nwalkers, b.nleaves_max, b.ndim):
                raise ValueError("incompatible input dimensions")
        if st.betas is not None:
            tc.betas = np.array(st.betas, dtype=np.float64)
        if st.log_like is None or st.log_prior is None:
            L, P = self._eval(coords, inds)
            st.log_like = L if st.log_like is None else st.log_like
            st.log_prior = P if st.log_prior is None else st.log_prior
        if np.any(np.isinf(st.log_prior)):
            raise ValueError("The initial log_prior was +/- infinite")
        self.engine.upload(coords, inds, st.log_like, st.log_prior, tc.betas)
        self.engine.set_adapt_time(tc.time)
        for phase, n, keep in (("burn", burn or 0, False), ("run", nsteps, store)):
            for _ in range(n):
                if self.rng == "philox":
                    self.engine.step(thin_by if phase == "run" else 1)
                else:
                    for _ in range(thin_by if phase == "run" else 1):
                        self._iteration_numpy()
                if keep:
                    self.chain.append(self._state(nan_fill=True))
                self.iteration += 1
        out = self._state()
        tc.betas = out.betas
        if self.rng == "philox":
            c = self.engine.counters()
            tc.time, tc.swaps_accepted = c["adapt_time"], c["swaps_last"]
            # the moves' own counters (move.py:404-421), cumulative over the context's life like the device's
            mv = self.moves[0]
            mv.accepted, mv.num_proposals = c["accepted_mh"].copy(), c["num_mh"]
            self.rj_accepted_all, self.rj_num_proposals_all = c["accepted_bd"].copy(), c["num_bd"]
        self._previous_state = out
        return out

    def get_nleaves(self):
        return {b.name: np.stack([s.branches[b.name].nleaves for s in self.chain]) for b in self.branches}
