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


class _TemplateLikelihood:
    kind = _lib.LIKE_TEMPLATE

    def __init__(self, ndim):
        self.ndim = ndim

    def _install(self, lib, ctx):
        pass


class RJEngine:
    def __init__(self, ntemps, nwalkers, branches, t, y, sigma, seed=0, device_id=0, adaptive=True,
                 adaptation_lag=10000, adaptation_time=100, stop_adaptation=-1, fill_value=-1e300):
        self.branches = list(branches)
        if not 1 <= len(self.branches) <= 4:
            raise NotImplementedError("1 to 4 branches")
        self.ncoord = sum(b.nleaves_max * 3 for b in self.branches)
        rw = self.ncoord + len(self.branches)
        self.RW = rw + (rw & 1)                                  # even record width: 16-byte row alignment
        self.off = np.cumsum([0] + [b.nleaves_max * 3 for b in self.branches])[:-1]
        self.T, self.W = int(ntemps), int(nwalkers)
        # the engine's prior box is unused on records; HipEnsemble wants one
        self.eng = HipEnsemble(self.T, self.W, self.RW, _TemplateLikelihood(self.RW), -1.0, 1.0, tempered=True,
                               adaptive=adaptive, adaptation_lag=adaptation_lag, adaptation_time=adaptation_time,
                               stop_adaptation=stop_adaptation, live_dangerously=True, fill_value=fill_value, seed=seed,
                               device_id=device_id)
        self.lib, self.ctx = self.eng.lib, self.eng.ctx
        nb = len(self.branches)
        kinds = np.array([b.kind for b in self.branches], dtype=np.int32)
        nlmax = np.array([b.nleaves_max for b in self.branches], dtype=np.int32)
        nlmin = np.array([b.nleaves_min for b in self.branches], dtype=np.int32)
        lo = f64(np.stack([b.lo for b in self.branches]))
        hi = f64(np.stack([b.hi for b in self.branches]))
        lp = f64([b.leaf_logp for b in self.branches])
        t, y = f64(t), f64(y)
        if t.shape != y.shape or t.ndim != 1:
            raise ValueError("t and y must be 1-D arrays of the same length")
        check(self.lib.hens_rj_set_model(self.ctx, nb, ptr(kinds), ptr(nlmax), ptr(nlmin), ptr(lo), ptr(hi), ptr(lp),
                                         int(t.shape[0]), ptr(t), ptr(y), float(sigma)), self.ctx)

    def close(self):
        self.eng.close()

    # -- records <-> branches --------------------------------------------------------------------------------
    def pack(self, x, inds):
        """{name: coords[T, W, nl, 3]}, {name: inds[T, W, nl]} -> records[T, W, RW]."""
        rec = np.zeros((self.T, self.W, self.RW))
        for bi, b in enumerate(self.branches):
            c = np.asarray(x[b.name], dtype=np.float64)
            if c.shape != (self.T, self.W, b.nleaves_max, 3):
                raise ValueError(f"coords of branch {b.name} must have shape {(self.T, self.W, b.nleaves_max, 3)}")
            c = np.where(np.isnan(c), 0.0, c)                    # a stored chain marks unused leaves with NaN
            rec[:, :, self.off[bi]:self.off[bi] + b.nleaves_max * 3] = c.reshape(self.T, self.W, -1)
            m = np.asarray(inds[b.name], dtype=bool)
            rec[:, :, self.ncoord + bi] = (m * (1 << np.arange(b.nleaves_max))).sum(axis=-1)
        return rec

    def unpack(self, rec, nan_fill=False):
        x, inds = {}, {}
        for bi, b in enumerate(self.branches):
            x[b.name] = rec[:, :, self.off[bi]:self.off[bi] + b.nleaves_max * 3].reshape(self.T, self.W, b.nleaves_max, 3).copy()
            m = rec[:, :, self.ncoord + bi].astype(np.int64)
            inds[b.name] = ((m[:, :, None] >> np.arange(b.nleaves_max)) & 1).astype(bool)
            if nan_fill:                                         # backend.py:1049-1059
                x[b.name][~inds[b.name]] = np.nan
        return x, inds

    def steps_to_records(self, steps):
        """{name: step[T, W, nl, 3]} (zero on unused slots) -> [T, W, ncoord] in record layout."""
        out = np.zeros((self.T, self.W, self.ncoord))
        for bi, b in enumerate(self.branches):
            out[:, :, self.off[bi]:self.off[bi] + b.nleaves_max * 3] = np.asarray(steps[b.name]).reshape(self.T, self.W, -1)
        return out

    # -- state ---------------------------------------------------------------------------------------------------
    def upload(self, x, inds, logl=None, logp=None, betas=None):
        self.eng.upload(self.pack(x, inds), logl, logp, betas)

    def download(self, nan_fill=False):
        rec, L, P, betas = self.eng.download()
        x, inds = self.unpack(rec, nan_fill=nan_fill)
        return x, inds, L, P, betas

    def eval_state(self):
        self.eng.eval_state()

    def set_adapt_time(self, t):
        self.eng.set_adapt_time(t)

    # -- parity-mode moves -----------------------------------------------------------------------------------------
    def mh_step(self, steps, u_acc):
        st = f64(self.steps_to_records(steps))
        u = f64(u_acc, (self.T, self.W))
        keep = np.empty((self.T, self.W), dtype=np.uint8)
        check(self.lib.hens_rj_mh_step(self.ctx, ptr(st), ptr(u), ptr(keep)), self.ctx)
        return keep.astype(bool)

    def bd_step(self, branch, change, leaf, birth, u_acc):
        """change[T, W] in {-1, 0, +1}, leaf[T, W] slot, birth[T, W, 3] (rows of walkers that give birth), u_acc[T, W]."""
        ch = np.ascontiguousarray(change, dtype=np.int8)
        lf = np.ascontiguousarray(np.where(np.asarray(change) == 0, 0, leaf), dtype=np.int32)
        bt = f64(birth, (self.T, self.W, 3))
        u = f64(u_acc, (self.T, self.W))
        keep = np.empty((self.T, self.W), dtype=np.uint8)
        check(self.lib.hens_rj_bd_step(self.ctx, int(branch), ptr(ch), ptr(lf), ptr(bt), ptr(u), ptr(keep)), self.ctx)
        return keep.astype(bool)

    def pt_sweep(self, iperm, i1perm, u_swap, adapt=True):
        return self.eng.pt_sweep(iperm, i1perm, u_swap, adapt=adapt)

    # -- production -------------------------------------------------------------------------------------------------
    def set_mh_scale(self, scale):
        """Standard deviations of the in-model Gaussian step per branch and leaf parameter: [nbranches, 3]."""
        s = f64(scale, (len(self.branches), 3))
        check(self.lib.hens_rj_set_mh_scale(self.ctx, ptr(s)), self.ctx)

    def step(self, n_iters):
        check(self.lib.hens_rj_step(self.ctx, int(n_iters)), self.ctx)

    def synchronize(self):
        self.eng.synchronize()

    def counters(self):
        c = self.eng.counters()
        bd = np.empty((self.T, self.W))
        n_mh, n_bd = C.c_int64(0), C.c_int64(0)
        check(self.lib.hens_rj_get_counters(self.ctx, ptr(bd), C.byref(n_mh), C.byref(n_bd)), self.ctx)
        c.update(accepted_mh=c["accepted"], accepted_bd=bd, num_mh=int(n_mh.value), num_bd=int(n_bd.value))
        return c
