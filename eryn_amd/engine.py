"""HipEnsemble: thin object wrapper over the C ABI (one context = one GPU = one ladder shard)."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import HensConfig, HensDeviceBuffers, HensTiming, check, f64, ptr


def box_logp_inside(lo, hi):
    """sum_d log(1/(hi_d - lo_d)) accumulated in the reference's order
    (prior.py:28-41 ``logpdf_val``; prior.py:364-383 sequential ``prior_vals += temp``)."""
    acc = np.zeros(1)
    for d in range(len(lo)):
        acc += np.log(1 / (hi[d] - lo[d]))
    return float(acc[0])


FAST_WIDTHS = (8, 16, 32, 64, 128)       # row widths with compile-time-width kernels (csrc/hens.hip: fast_path)


def padded_width(ndim, likelihood):
    """Row width of the device pool for ``ndim`` real parameters.

    The compile-time-width kernels (in-place two-launch / one-launch iterations) serve rows of 8, 16, 32, 64 and 128
    doubles; other widths run the generic kernel in three copying launches (measured at 16 x 4096: D = 11 39.1 us,
    D = 12 34.5 us, D = 24 46.1 us per iteration against 16.1 us at D = 16 and 22.4 us at D = 32).  So rows of a Gaussian
    likelihood are padded up to the next such width with coordinates that never change: zeros, inside a (-inf, +inf) prior
    interval, meeting zero rows / columns of the precision matrix (``hens_config::ndim_active`` keeps the Hastings factor
    and the walker-count guard on the real dimension).  Not for the Rosenbrock likelihood (its coupled sum would see the
    pad) and host-callable likelihoods (their proposals travel to the host unpadded).  A property of (ndim, likelihood
    kind) only: every rank of a sharded ladder pads alike, so shards stay bit-identical to the whole ladder and agree on
    the row width of their messages.  ``HENS_NO_PAD=1`` switches it off."""
    if os.environ.get("HENS_NO_PAD") or ndim in FAST_WIDTHS or ndim > FAST_WIDTHS[-1]:
        return ndim
    if likelihood.kind not in (_lib.LIKE_GAUSS_DENSE, _lib.LIKE_GAUSS_DIAG):
        return ndim
    return next(w for w in FAST_WIDTHS if w >= ndim)


class HipEnsemble:
    def __init__(self, ntemps, nwalkers, ndim, likelihood, lo, hi, a=2.0, tempered=None,
                 adaptive=True, adaptation_lag=10000, adaptation_time=100, stop_adaptation=-1,
                 live_dangerously=False, fill_value=-1e300, seed=0, rung_range=None, device_id=0,
                 adaptation_delay=0, pad_rows=True):
        self.lib = _lib.load()
        self.T, self.W, self.D = int(ntemps), int(nwalkers), int(ndim)
        if tempered is None:
            tempered = self.T > 1
        r0, r1 = (0, self.T) if rung_range is None else (int(rung_range[0]), int(rung_range[1]))
        self.rung_begin, self.rung_end, self.Tl = r0, r1, r1 - r0
        if likelihood.ndim != self.D:
            raise ValueError("likelihood dimension does not match ndim")
        # RW: width of a device row (>= D, see padded_width); every array with a parameter axis is padded on the way in and
        # stripped on the way out, the caller only ever sees D
        self.RW = padded_width(self.D, likelihood) if pad_rows else self.D
        cfg = HensConfig(ntemps=self.T, nwalkers=self.W, ndim=self.RW, ndim_active=self.D if self.RW != self.D else 0,
                         rung_begin=r0, rung_end=r1,
                         device_id=int(device_id), likelihood_kind=int(likelihood.kind),
                         tempered=int(bool(tempered)), live_dangerously=int(bool(live_dangerously)),
                         adaptive=int(bool(adaptive)), adaptation_delay=int(adaptation_delay),
                         stop_adaptation=int(stop_adaptation), a=float(a),
                         fill_value=float(fill_value), adaptation_lag=float(adaptation_lag),
                         adaptation_time=float(adaptation_time), seed=int(seed) & (2**64 - 1))
        self.tempered = bool(tempered)
        self.state_epoch = 0         # bumped by every call that changes the walkers: a DeviceState (eryn_amd/state.py) is current while it matches
        self.lazy_downloads = 0
        self.a = float(a)
        self.ctx = C.c_void_p()
        code = self.lib.hens_create(C.byref(cfg), C.byref(self.ctx))
        if code != _lib.HENS_OK:
            self.ctx = None
            check(code, None)
        self.lo = f64(np.broadcast_to(lo, (self.D,)))
        self.hi = f64(np.broadcast_to(hi, (self.D,)))
        self.logp_inside = box_logp_inside(self.lo, self.hi)
        lo_w, hi_w = self._pad(self.lo, -np.inf), self._pad(self.hi, np.inf)
        check(self.lib.hens_set_prior_box(self.ctx, ptr(lo_w), ptr(hi_w), self.logp_inside), self.ctx)
        if self.RW != self.D:
            likelihood._install(self.lib, self.ctx, row_width=self.RW)
        else:
            likelihood._install(self.lib, self.ctx)
        self.likelihood = likelihood
        self.N0 = (self.W + 1) // 2

    def _pad(self, a, fill=0.0):
        """``a[..., D]`` -> C-contiguous ``[..., RW]`` with ``fill`` on the pads (the array itself when nothing is padded)."""
        if self.RW == self.D:
            return a
        out = np.full(a.shape[:-1] + (self.RW,), fill, dtype=np.float64)
        out[..., :self.D] = a
        return out

    def set_stretch_scale(self, a):
        """``StretchMove.a`` for every later proposal (the reference's tuning hook mutates it, utils/updates.py:130-175)."""
        check(self.lib.hens_set_stretch_scale(self.ctx, float(a)), self.ctx)
        self.a = float(a)

    def set_periodic(self, period):
        """Periods of the periodic parameters, ``[ndim]`` (0 = not periodic), or None for none: the ``periodic``
        argument of the reference's sampler / moves for the single branch (ensemble.py:165-168, utils/periodic.py)."""
        if period is None:
            check(self.lib.hens_set_periodic(self.ctx, None), self.ctx)
            self.period = None
            return
        self.period = f64(period, (self.D,))
        check(self.lib.hens_set_periodic(self.ctx, ptr(self._pad(self.period))), self.ctx)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.hens_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state ---------------------------------------------------------------------------------
    def upload(self, x, logl=None, logp=None, betas=None):
        self.state_epoch += 1
        x = self._pad(f64(x, (self.Tl, self.W, self.D)))
        logl = None if logl is None else f64(logl, (self.Tl, self.W))
        logp = None if logp is None else f64(logp, (self.Tl, self.W))
        betas = None if betas is None else f64(betas, (self.T,))
        check(self.lib.hens_upload_state(self.ctx, ptr(x), ptr(logl), ptr(logp), ptr(betas)), self.ctx)

    def download(self, want_x=True):
        x = np.empty((self.Tl, self.W, self.RW)) if want_x else None
        logl = np.empty((self.Tl, self.W))
        logp = np.empty((self.Tl, self.W))
        betas = np.empty(self.T) if self.tempered else None
        check(self.lib.hens_download_state(self.ctx, ptr(x), ptr(logl), ptr(logp), ptr(betas)), self.ctx)
        if want_x and self.RW != self.D:
            x = np.ascontiguousarray(x[..., :self.D])
        return x, logl, logp, betas

    def eval_state(self):
        self.state_epoch += 1
        check(self.lib.hens_eval_state(self.ctx), self.ctx)

    # -- parity-mode steps -----------------------------------------------------------------------
    nsplits = 2

    def set_nsplits(self, nsplits):
        """Sets of the parity API's red-blue move (``RedBlueMove(nsplits=...)``, red_blue.py:41-47): include/hipensemble.h,
        hens_set_nsplits."""
        check(self.lib.hens_set_nsplits(self.ctx, int(nsplits)), self.ctx)
        self.nsplits = int(nsplits)

    def set_size(self, split):
        """Walkers of set ``split``: arange(W) % nsplits, shuffled (red_blue.py:119-124)."""
        return (self.W - int(split) + self.nsplits - 1) // self.nsplits

    def stretch_split(self, split, labels, rint, u_zz, u_acc):
        self.state_epoch += 1
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        if labels.shape != (self.Tl, self.W):
            raise ValueError("labels must have shape (ntemps, nwalkers)")
        Ns = self.set_size(split)
        rint = np.ascontiguousarray(rint, dtype=np.int64)
        u_zz, u_acc = f64(u_zz, (self.Tl, Ns)), f64(u_acc, (self.Tl, Ns))
        if rint.shape != (self.Tl, Ns):
            raise ValueError(f"rint must have shape {(self.Tl, Ns)}")
        keep = np.empty((self.Tl, Ns), dtype=np.uint8)
        check(self.lib.hens_stretch_split(self.ctx, int(split), ptr(labels), ptr(rint), ptr(u_zz), ptr(u_acc),
                                          ptr(keep)), self.ctx)
        return keep.astype(bool)

    def propose_split(self, split, labels, rint, u_zz):
        """Host-likelihood contexts: proposed points q[Tl, Ns, D] and the in-prior mask [Tl, Ns]."""
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        Ns = self.set_size(split)
        rint = np.ascontiguousarray(rint, dtype=np.int64)
        u_zz = f64(u_zz, (self.Tl, Ns))
        if labels.shape != (self.Tl, self.W) or rint.shape != (self.Tl, Ns):
            raise ValueError("labels / rint have the wrong shape")
        q = np.empty((self.Tl, Ns, self.D))
        inbox = np.empty((self.Tl, Ns), dtype=np.uint8)
        check(self.lib.hens_propose_split(self.ctx, int(split), ptr(labels), ptr(rint), ptr(u_zz), ptr(q),
                                          ptr(inbox)), self.ctx)
        return q, inbox.astype(bool)

    def accept_split(self, split, logl, u_acc):
        self.state_epoch += 1
        Ns = self.set_size(split)
        logl, u_acc = f64(logl, (self.Tl, Ns)), f64(u_acc, (self.Tl, Ns))
        keep = np.empty((self.Tl, Ns), dtype=np.uint8)
        check(self.lib.hens_accept_split(self.ctx, int(split), ptr(logl), ptr(u_acc), ptr(keep)), self.ctx)
        return keep.astype(bool)

    def pt_sweep(self, iperm, i1perm, u_swap, adapt=True):
        self.state_epoch += 1
        shp = (self.T - 1, self.W)
        iperm = np.ascontiguousarray(iperm, dtype=np.int64)
        i1perm = np.ascontiguousarray(i1perm, dtype=np.int64)
        u_swap = f64(u_swap, shp)
        if iperm.shape != shp or i1perm.shape != shp:
            raise ValueError(f"iperm/i1perm must have shape {shp}")
        sel = np.zeros(shp, dtype=np.uint8)
        swaps = np.zeros(max(self.T - 1, 0))
        check(self.lib.hens_pt_sweep(self.ctx, ptr(iperm), ptr(i1perm), ptr(u_swap), int(bool(adapt)), ptr(sel),
                                     ptr(swaps)), self.ctx)
        return sel.astype(bool), swaps

    # -- production ------------------------------------------------------------------------------
    def step(self, n_iters):
        self.state_epoch += 1
        check(self.lib.hens_step(self.ctx, int(n_iters)), self.ctx)

    def step_marked(self, n_before, n_last):
        """n_before iterations, the accept counters kept on the device, n_last iterations (thin_by > 1: the reference stores
        the accept mask of the last sub-iteration only, ensemble.py:968-979); see marked_counters."""
        self.state_epoch += 1
        check(self.lib.hens_step_marked(self.ctx, int(n_before), int(n_last)), self.ctx)

    def step_report(self, n_iters, n_last=1):
        """``step(n_iters)`` + what a sampler loop reads after every proposal (ensemble.py:974-977) in one call: the accept counts
        of the last ``n_last`` iterations ``[Tl, W]`` (uint8), the last cascade's swap counts, the ladder.  The walkers stay on
        the device (include/hipensemble.h: hens_step_report)."""
        self.state_epoch += 1
        acc = np.empty((self.Tl, self.W), dtype=np.uint8)
        swaps = np.zeros(max(self.T - 1, 0))
        betas = np.empty(self.T) if self.tempered else None
        check(self.lib.hens_step_report(self.ctx, int(n_iters), int(n_last), ptr(acc), ptr(swaps) if self.T > 1 else None, ptr(betas)), self.ctx)
        return acc, swaps, betas

    def download_betas(self):
        betas = np.empty(self.T)
        check(self.lib.hens_download_state(self.ctx, None, None, None, ptr(betas)), self.ctx)
        return betas

    def marked_counters(self):
        """The accept counts (stretch move, MH move) in front of the last step_marked call's final iterations."""
        acc = np.zeros((self.Tl, self.W))
        acc_mh = np.zeros((self.Tl, self.W))
        check(self.lib.hens_get_marked_counters(self.ctx, ptr(acc), ptr(acc_mh)), self.ctx)
        return acc, acc_mh

    def synchronize(self):
        check(self.lib.hens_synchronize(self.ctx), self.ctx)

    def counters(self):
        acc = np.empty((self.Tl, self.W))
        nprop, atime = C.c_int64(0), C.c_int64(0)
        last = np.zeros(max(self.T - 1, 0))
        total = np.zeros(max(self.T - 1, 0))
        check(self.lib.hens_get_counters(self.ctx, ptr(acc), C.byref(nprop), ptr(last), ptr(total),
                                         C.byref(atime)), self.ctx)
        return dict(accepted=acc, num_proposals=nprop.value, swaps_last=last, swaps_total=total,
                    adapt_time=atime.value)

    def reset_counters(self):
        check(self.lib.hens_reset_counters(self.ctx), self.ctx)

    def set_adapt_time(self, t):
        check(self.lib.hens_set_adapt_time(self.ctx, int(t)), self.ctx)

    def debug_permutation(self, which, rung, it):
        out = np.empty(self.W, dtype=np.int32)
        check(self.lib.hens_debug_permutation(self.ctx, int(which), int(rung), int(it), ptr(out)), self.ctx)
        return out

    def set_iteration(self, it):
        """Resume: move the Philox iteration counter (hens_set_iteration) - with the same seed, an uploaded state and the
        adaptation time restored, ``step`` continues a stored chain bit for bit."""
        check(self.lib.hens_set_iteration(self.ctx, int(it)), self.ctx)

    def iteration(self):
        """Index of the next Philox iteration ``step`` will run."""
        n = C.c_int64(0)
        check(self.lib.hens_get_iteration(self.ctx, C.byref(n)), self.ctx)
        return int(n.value)

    def debug_draws(self, it, mh=False):
        """The Philox draws ``step`` consumes in iteration ``it`` (include/hipensemble.h: hens_debug_draws)."""
        T, Tl, W, D = self.T, self.Tl, self.W, self.D
        out = dict(own=np.empty((Tl, W), dtype=np.int32), cw=np.empty((Tl, W), dtype=np.int32),
                   u_zz=np.empty((Tl, W)), u_acc=np.empty((Tl, W)))
        pt = self.tempered and T > 1
        if pt:
            out.update(pt_slot=np.empty((T, W), dtype=np.int32), u_swap=np.empty((T - 1, W)))
        if mh:
            out.update(mh_step=np.empty((Tl, W, self.RW)), mh_u=np.empty((Tl, W)))
        is_mh = C.c_int32(0)
        check(self.lib.hens_debug_draws(self.ctx, int(it), ptr(out["own"]), ptr(out["cw"]), ptr(out["u_zz"]),
                                        ptr(out["u_acc"]), ptr(out.get("pt_slot")), ptr(out.get("u_swap")),
                                        C.byref(is_mh), ptr(out.get("mh_step")), ptr(out.get("mh_u"))), self.ctx)
        out["is_mh"] = bool(is_mh.value)
        if mh and self.RW != D:
            out["mh_step"] = np.ascontiguousarray(out["mh_step"][..., :D])
        return out

    def set_profiling(self, on):
        """0 / False off; 1 / True a HIP event pair per launch (HIP stream); 2 the launches' own dispatch timestamps on the queue
        the call uses anyway (include/hipensemble.h: hens_set_profiling)."""
        check(self.lib.hens_set_profiling(self.ctx, int(on)), self.ctx)

    def timing(self):
        t = HensTiming()
        check(self.lib.hens_get_timing(self.ctx, C.byref(t)), self.ctx)
        return {k: getattr(t, k) for k, _ in HensTiming._fields_}

    def launch_times(self):
        """(n_launches, 2) begin / end in us after the first launch's begin, of the last step() made with set_profiling(True)."""
        n = C.c_int64(0)
        check(self.lib.hens_debug_launch_times(self.ctx, None, 0, C.byref(n)), self.ctx)
        out = np.zeros(max(n.value, 0))
        if n.value:
            check(self.lib.hens_debug_launch_times(self.ctx, ptr(out), n.value, C.byref(n)), self.ctx)
        return out.reshape(-1, 2)

    # -- ladder sharding (eryn_amd/ladder.py) ----------------------------------------------------------
    def stretch_iter(self):
        """One Philox iteration of both halves on the resident rungs (asynchronous, no PT)."""
        self.state_epoch += 1
        check(self.lib.hens_stretch_iter(self.ctx), self.ctx)

    def device_buffers(self):
        b = HensDeviceBuffers()
        check(self.lib.hens_get_device_buffers(self.ctx, C.byref(b)), self.ctx)
        return b

    def pt_plan_sharded(self, rank_of_rung, nranks, my_rank, iperm=None, i1perm=None, u_swap=None, adapt=True,
                        want_swaps=True):
        rank_of_rung = np.ascontiguousarray(rank_of_rung, dtype=np.int32)
        if rank_of_rung.shape != (self.T,):
            raise ValueError("rank_of_rung must have shape (ntemps,)")
        shp = (self.T - 1, self.W)
        if iperm is not None:
            iperm = np.ascontiguousarray(iperm, dtype=np.int64)
            i1perm = np.ascontiguousarray(i1perm, dtype=np.int64)
            u_swap = f64(u_swap, shp)
            if iperm.shape != shp or i1perm.shape != shp:
                raise ValueError(f"iperm/i1perm must have shape {shp}")
        send = np.zeros(nranks, dtype=np.int64)
        recv = np.zeros(nranks, dtype=np.int64)
        sel = np.zeros(shp, dtype=np.uint8) if iperm is not None else None
        swaps = np.zeros(self.T - 1) if want_swaps else None
        check(self.lib.hens_pt_plan_sharded(self.ctx, ptr(iperm), ptr(i1perm), ptr(u_swap), int(bool(adapt)),
                                            ptr(rank_of_rung), int(nranks), int(my_rank), ptr(send), ptr(recv),
                                            ptr(sel), ptr(swaps)), self.ctx)
        return send, recv, (None if sel is None else sel.astype(bool)), swaps

    def set_stream(self, stream_handle):
        """Launch on a caller-owned HIP stream (e.g. torch's current stream, so RCCL collectives and the
        kernels order themselves without host synchronisation)."""
        check(self.lib.hens_set_stream(self.ctx, C.c_void_p(int(stream_handle))), self.ctx)

    # -- Metropolis-Hastings proposals (include/hipensemble.h: hens_mh_*) --------------------------------
    def mh_step(self, step, u_acc):
        """One full-ensemble proposal q = x + step with the caller's draws; returns the accept mask [Tl, W]."""
        self.state_epoch += 1
        step = self._pad(np.ascontiguousarray(step, dtype=np.float64).reshape(self.Tl, self.W, self.D))
        u_acc = np.ascontiguousarray(u_acc, dtype=np.float64).reshape(self.Tl, self.W)
        keep = np.empty((self.Tl, self.W), dtype=np.uint8)
        check(self.lib.hens_mh_step(self.ctx, ptr(step), ptr(u_acc), ptr(keep)), self.ctx)
        return keep.astype(bool)

    def set_mh_proposal(self, kind, scale, weight):
        """Mix Gaussian MH proposals into ``step()``: kind "iso" | "diag" | "full" (scale = std dev, std devs,
        lower Cholesky factor), weight = probability per iteration; kind None switches the mix off."""
        self._move_cfg = None                                  # (DeviceMove._propose_philox's memo of what it asked for last)
        if kind is None:
            check(self.lib.hens_set_mh_proposal(self.ctx, -1, None, 0.0), self.ctx)
            return
        k = {"iso": 0, "diag": 1, "full": 2}[kind]
        scale = np.ascontiguousarray(np.atleast_1d(scale), dtype=np.float64)
        need = {0: 1, 1: self.D, 2: self.D * self.D}[k]
        if scale.size != need:
            raise ValueError(f"{kind} proposal needs {need} scale value(s)")
        if self.RW != self.D and k == 0:                       # no step on the pads: the same normals, per-parameter scales
            k, scale = 1, self._pad(np.full(self.D, float(scale[0])))
        elif self.RW != self.D and k == 1:
            scale = self._pad(scale)
        elif self.RW != self.D and k == 2:
            full = np.zeros((self.RW, self.RW))
            full[:self.D, :self.D] = scale.reshape(self.D, self.D)
            scale = full
        check(self.lib.hens_set_mh_proposal(self.ctx, k, ptr(scale), float(weight)), self.ctx)

    def mh_counters(self):
        acc = np.zeros((self.Tl, self.W))
        n = C.c_int64(0)
        check(self.lib.hens_get_mh_counters(self.ctx, ptr(acc), C.byref(n)), self.ctx)
        return dict(accepted=acc, num_proposals=int(n.value))

    # -- RCCL neighbour exchange inside the library (include/hipensemble.h: hens_comm_*) ----------------
    def comm_unique_id(self):
        """128 bytes from ncclGetUniqueId (call on ONE rank, hand the bytes to the others)."""
        buf = (C.c_ubyte * 128)()
        check(self.lib.hens_comm_unique_id(C.cast(buf, C.c_void_p)), None)
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id):
        """Collective: this shard becomes rank `rank` of `nranks`; ``step(n)`` then sends the neighbour messages itself."""
        if len(unique_id) != 128:
            raise ValueError("unique_id: 128 bytes from comm_unique_id()")
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        check(self.lib.hens_comm_init(self.ctx, int(nranks), int(rank), C.cast(buf, C.c_void_p)), self.ctx)

    def comm_destroy(self):
        check(self.lib.hens_comm_destroy(self.ctx), self.ctx)

    def comm_selfsend(self, values):
        src = np.ascontiguousarray(values, dtype=np.float64)
        dst = np.empty_like(src)
        check(self.lib.hens_comm_selfsend(self.ctx, src.size, ptr(src), ptr(dst)), self.ctx)
        return dst

    # -- ladder pipeline (include/hipensemble.h: hens_pipe_*) ------------------------------------------
    def pipe_init(self, nranks, rank):
        """Allocate this shard's mailbox; returns the 128-byte blob of HIP IPC handles (mailbox, walker pool)."""
        h = (C.c_ubyte * 128)()
        nbytes = C.c_int64(0)
        check(self.lib.hens_pipe_init(self.ctx, int(nranks), int(rank), C.cast(h, C.c_void_p), C.byref(nbytes)), self.ctx)
        self.pipe_bytes = int(nbytes.value)
        return bytes(h)

    def pipe_connect(self, handles):
        """handles: the ranks' IPC blobs concatenated in rank order (other processes)."""
        buf = (C.c_ubyte * len(handles)).from_buffer_copy(handles)
        check(self.lib.hens_pipe_connect(self.ctx, C.cast(buf, C.c_void_p)), self.ctx)

    def pipe_connect_local(self, engines):
        """All shards live in this process (several contexts on one GPU)."""
        arr = (C.c_void_p * len(engines))(*[e.ctx for e in engines])
        check(self.lib.hens_pipe_connect_local(self.ctx, C.cast(arr, C.c_void_p)), self.ctx)

    def pipe_connect_staged(self):
        """Staged transport: the pipeline's kernels write into local outboxes, the caller moves the messages."""
        check(self.lib.hens_pipe_connect_staged(self.ctx), self.ctx)

    def pipe_regions(self):
        from ._lib import HensPipeRegions
        r = HensPipeRegions()
        check(self.lib.hens_pipe_regions(self.ctx, C.byref(r)), self.ctx)
        return r

    def pipe_stage(self, stage):
        check(self.lib.hens_pipe_stage(self.ctx, int(stage)), self.ctx)

    def pipe_debug_stats(self, reset=True):
        """{site: (seconds waited summed over workgroups, waits)} - needs HENS_PIPE_STATS=1 at pipe_init."""
        raw = np.zeros(16, dtype=np.uint64)
        check(self.lib.hens_pipe_debug_stats(self.ctx, ptr(raw), int(bool(reset))), self.ctx)
        names = ["stretch:rows", "stretch:counts", "walk:columns", "bottom:cold rung", "bottom:rows from above",
                 "walk:collector waits for the grid", "walk:collector tail"]
        return {n: (float(raw[2 * i]) * 1e-8, int(raw[2 * i + 1])) for i, n in enumerate(names)}

    def pt_finish_sharded(self, n_recv):
        self.state_epoch += 1
        check(self.lib.hens_pt_finish_sharded(self.ctx, int(n_recv)), self.ctx)
