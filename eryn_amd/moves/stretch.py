"""HIP-backed ``StretchMove`` with the reference's plugin contract.

``propose(model, state) -> (state, accepted)`` is what ``EnsembleSampler.sample`` calls
(ensemble.py:974).  This class replaces, for the single-branch ``nleaves_max == 1`` case,
``RedBlueMove.propose`` (red_blue.py:89-333), ``StretchMove.get_proposal``
(stretch.py:160-231), the ``compute_log_prior`` / ``compute_log_like`` round trips, the
accept test, ``Move.update`` (move.py:472-703) and ``TemperatureControl.temper_comps``
(tempering.py:598-649) with calls into libhipensemble.  It draws from the two reference
streams (``model.random`` and the global ``np.random``) in the reference's order, so with
the same seeds it reproduces Eryn's chain (bit-exact masks and positions; log-likelihood to
1e-12 relative).
"""
import numpy as np

from ..engine import HipEnsemble
from ..state import State
from .move import Move

__all__ = ["StretchMove"]


class StretchMove(Move):
    """Goodman-Weare stretch move evaluated on the MI355X.

    Args beyond the reference's ``StretchMove(a=2.0, nsplits=2, randomize_split=True,
    live_dangerously=False)``:
        likelihood: a :mod:`eryn_amd.likelihood` object (the likelihood lives inside the kernel).
        prior_box: ``(lo, hi)`` scalars or ``[ndim]`` arrays, or a ``ProbDistContainer``.
        device_id, fill_value: engine options.
        trust_resident: skip the host->device upload when ``state`` is the object this move
            returned last (safe when no host-side code mutates the State in between).
    """

    def __init__(self, a=2.0, nsplits=2, randomize_split=True, live_dangerously=False, likelihood=None,
                 prior_box=None, device_id=0, fill_value=-1e300, trust_resident=False, return_gpu=False,
                 **kwargs):
        if nsplits != 2:
            raise NotImplementedError("the device path implements nsplits = 2")
        self.a = a
        self.nsplits = nsplits
        self.randomize_split = randomize_split
        self.live_dangerously = live_dangerously
        self.likelihood = likelihood
        self.prior_box = prior_box
        self.device_id = device_id
        self.fill_value = fill_value
        self.trust_resident = trust_resident
        self.engine = None
        self._resident = None
        Move.__init__(self, **kwargs)

    # -- engine -----------------------------------------------------------------------------------
    def _box(self):
        pb = self.prior_box
        if pb is None:
            raise ValueError("StretchMove needs prior_box=(lo, hi) or a ProbDistContainer")
        if hasattr(pb, "box_bounds"):
            return pb.box_bounds()
        return pb

    def attach_engine(self, engine):
        """Share one device context between the sampler, its moves and the temperature control."""
        self.engine = engine

    def _ensure_engine(self, T, W, D):
        e = self.engine
        if e is not None and (e.T, e.W, e.D) == (T, W, D):
            return e
        if self.likelihood is None:
            raise ValueError("StretchMove needs likelihood=<eryn_amd.likelihood object>")
        lo, hi = self._box()
        tc = self.temperature_control
        kw = {}
        if tc is not None:
            kw = dict(adaptive=tc.adaptive, adaptation_lag=tc.adaptation_lag, adaptation_time=tc.adaptation_time,
                      stop_adaptation=tc.stop_adaptation)
        self.engine = HipEnsemble(T, W, D, self.likelihood, lo, hi, a=self.a, tempered=tc is not None,
                                  live_dangerously=self.live_dangerously, fill_value=self.fill_value,
                                  device_id=self.device_id, **kw)
        return self.engine

    @staticmethod
    def _single_branch(state):
        names = list(state.branches.keys())
        if len(names) != 1:
            raise NotImplementedError("the device path handles a single branch (SURVEY 8f-4: RJ is a later row)")
        br = state.branches[names[0]]
        T, W, nl, D = br.shape
        if nl != 1 or not np.all(br.inds):
            raise NotImplementedError("the device path handles nleaves_max == 1 with all leaves active")
        if state.blobs is not None or state.supplemental is not None:
            raise NotImplementedError("blobs / supplementals are outside the device hot path")
        return names[0], br, T, W, D

    # -- the plugin entry point -------------------------------------------------------------------
    def propose(self, model, state):
        name, br, T, W, D = self._single_branch(state)
        eng = self._ensure_engine(T, W, D)
        tc = self.temperature_control
        if not (self.trust_resident and self._resident is state):
            if state.log_like is None or state.log_prior is None:
                raise ValueError("state must carry log_like and log_prior")
            eng.upload(br.coords[:, :, 0, :], state.log_like, state.log_prior,
                       None if tc is None else tc.betas)
            if tc is not None:
                eng.set_adapt_time(tc.time)

        accepted = np.zeros((T, W), dtype=bool)
        labels = np.tile(np.arange(W), (T, 1)) % self.nsplits         # red_blue.py:119-124
        if self.randomize_split:
            [np.random.shuffle(row) for row in labels]
        tt = np.arange(T)[:, None]
        for split in range(self.nsplits):
            S = np.stack([np.flatnonzero(labels[t] == split) for t in range(T)])
            Ns = S.shape[1]
            Nc = W - Ns
            rint = model.random.randint(Nc, size=(T, Ns))              # stretch.py:93-99
            u_zz = model.random.rand(T, Ns)                            # stretch.py:129-132
            if hasattr(eng.likelihood, "evaluate"):
                # arbitrary Python likelihood: propose on the device, evaluate here, accept on the device
                q, inbox = eng.propose_split(split, labels, rint, u_zz)
                logl = eng.likelihood.evaluate(q, inbox)               # ensemble.py:1219-1545
                u_acc = model.random.rand(T, Ns)                       # red_blue.py:294 (drawn after the likelihood)
                keep = eng.accept_split(split, logl, u_acc)
            else:
                u_acc = model.random.rand(T, Ns)                       # red_blue.py:294
                keep = eng.stretch_split(split, labels, rint, u_zz, u_acc)
            accepted[tt, S] = keep
        if self._accepted is not None:
            self.accepted += accepted                                  # red_blue.py:326-327
        self.num_proposals += 1

        if tc is not None and T > 1:                                   # red_blue.py:330-331
            iperm, i1perm, u = _swap_draws(tc, T, W)
            do_adapt = bool(tc.adaptive)
            sel, swaps = eng.pt_sweep(iperm, i1perm, u, adapt=do_adapt)
            tc.swaps_accepted = swaps
            x, L, P, betas = eng.download()
            if do_adapt:
                tc.betas = betas
                tc.time += 1
        else:
            x, L, P, betas = eng.download()
            if tc is not None:
                tc.swaps_accepted = np.empty(0)
        out = State({name: x[:, :, None, :]}, inds={name: br.inds}, log_like=L, log_prior=P,
                    betas=None if tc is None else tc.betas, random_state=state.random_state)
        self._resident = out
        return out, accepted


def _swap_draws(tc, T, W):
    if hasattr(tc, "draw_swap_randoms"):
        return tc.draw_swap_randoms()
    # duck-typed reference TemperatureControl: same draw order (tempering.py:515-535)
    iperm = np.empty((T - 1, W), dtype=np.int64)
    i1perm = np.empty((T - 1, W), dtype=np.int64)
    u = np.empty((T - 1, W))
    for j in range(T - 1):
        if tc.permute:
            iperm[j] = np.random.permutation(W)
            i1perm[j] = np.random.permutation(W)
        else:
            iperm[j] = i1perm[j] = np.arange(W)
        u[j] = np.random.uniform(size=W)
    return iperm, i1perm, u
