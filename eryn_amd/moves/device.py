"""Plumbing shared by the HIP-backed moves: the device context, the resident-state shortcut and the
parallel-tempering tail every in-model move ends with (red_blue.py:330-331, mh.py:190-191)."""
import numpy as np

from ..engine import HipEnsemble
from ..periodic import period_vector
from ..state import DeviceState, State
from .move import Move

__all__ = ["DeviceMove"]


class DeviceMove(Move):
    """A move evaluated by libhipensemble.  Moves of one sampler share ONE device context
    (``attach_engine``) so the walkers stay resident between different moves of a weighted mix."""

    # the W >= 2 ndim guard is the red-blue move's (red_blue.py:108-114); other moves create their context without it
    needs_walker_guard = False

    def __init__(self, likelihood=None, prior_box=None, device_id=0, fill_value=-1e300, trust_resident=False,
                 a=2.0, live_dangerously=False, rng="numpy", lazy_state=None, seed=None, **kwargs):
        if rng not in ("numpy", "philox"):
            raise ValueError("rng must be 'numpy' or 'philox'")
        self.likelihood = likelihood
        self.prior_box = prior_box
        self.device_id = device_id
        self.fill_value = fill_value
        self.trust_resident = trust_resident
        # rng="numpy": the reference's two streams, drawn on the host in the reference's order and handed to the parity API - same
        #   seeds, same chain as Eryn.  rng="philox" (round 6): propose() IS one hens_step iteration - device-side draws, the
        #   production kernels - and returns what the sampler loop reads after every proposal (accept mask, swap counts, ladder).
        # lazy_state: propose() returns a DeviceState (eryn_amd/state.py) - the walkers are copied back when somebody reads them,
        #   not after every proposal.  Default: on with rng="philox", off with rng="numpy" (a State returned in the lazy form must
        #   be read before the next proposal if it is to be kept).
        self.rng = rng
        self.lazy_state = (rng == "philox") if lazy_state is None else bool(lazy_state)
        self.seed = seed
        self.engine = None
        self._resident = None
        self._engine_a = a
        self._engine_live = live_dangerously
        Move.__init__(self, **kwargs)

    # -- accept counts of the device-draw mode: folded into the sampler's array on demand ----------------------
    _pend, _npend = None, 0

    def _fold(self):
        if self._pend is not None and self._npend:
            self._accepted += self._pend
            self._pend[...] = 0
            self._npend = 0

    @property
    def accepted(self):
        self._fold()
        return Move.accepted.fget(self)

    @accepted.setter
    def accepted(self, accepted):
        if self._npend:                       # (a sampler replaces the array: what is pending belongs to the old one)
            self._fold()
        Move.accepted.fset(self, accepted)

    # -- engine -----------------------------------------------------------------------------------
    def _box(self):
        pb = self.prior_box
        if pb is None:
            raise ValueError(f"{type(self).__name__} needs prior_box=(lo, hi) or a ProbDistContainer")
        if hasattr(pb, "box_bounds"):
            return pb.box_bounds()
        return pb

    def attach_engine(self, engine, shared=None):
        """Share one device context between the sampler, its moves and the temperature control.
        ``shared``: a one-element list holding the State the context currently mirrors (so that the
        resident-state shortcut works across the moves of a mix)."""
        self.engine = engine
        self._shared = shared

    def _ensure_engine(self, T, W, D):
        e = self.engine
        if e is not None and (e.T, e.W, e.D) == (T, W, D):
            return e
        if self.likelihood is None:
            raise ValueError(f"{type(self).__name__} needs likelihood=<eryn_amd.likelihood object>")
        lo, hi = self._box()
        tc = self.temperature_control
        kw = {}
        if tc is not None:
            kw = dict(adaptive=tc.adaptive, adaptation_lag=tc.adaptation_lag, adaptation_time=tc.adaptation_time,
                      stop_adaptation=tc.stop_adaptation)
        seed = self.seed
        if seed is None:
            seed = int(np.random.randint(0, 2**31 - 1)) if self.rng == "philox" else 0
        self.engine = HipEnsemble(T, W, D, self.likelihood, lo, hi, a=self._engine_a, tempered=tc is not None,
                                  live_dangerously=self._engine_live or not self.needs_walker_guard, fill_value=self.fill_value,
                                  device_id=self.device_id, seed=seed, **kw)
        return self.engine

    def _apply_periodic(self, eng, name, D):
        """The context measures distances / wraps proposals with this move's periodic parameters (stretch.py:136-154,
        gaussian.py:110-115); moves of one sampler may differ, so every proposal states its own."""
        per = period_vector(self.periodic, name, D)
        cur = getattr(eng, "period", None)
        if (per is None) != (cur is None) or (per is not None and not np.array_equal(per, cur)):
            eng.set_periodic(per)

    @staticmethod
    def _single_branch(state):
        names = list(state.branches.keys())
        if len(names) != 1:
            raise NotImplementedError("this move steps a single-branch state; states of several branches / leaves step on leaf-packing "
                                      "records: eryn_amd.rj.RJEnsembleSampler with moves=StretchLeafMove() or GaussianLeafMove(cov)")
        br = state.branches[names[0]]
        T, W, nl, D = br.shape
        if isinstance(state, DeviceState):         # (made by a device move: one leaf, all active, no blobs)
            return names[0], br, T, W, D
        if nl != 1 or not np.all(br.inds):
            raise NotImplementedError("this move steps nleaves_max == 1 with all leaves active; several leaves step on leaf-packing "
                                      "records (eryn_amd.rj.RJEnsembleSampler)")
        if state.blobs is not None or state.supplemental is not None:
            raise NotImplementedError("blobs / supplementals are outside the device hot path")
        return names[0], br, T, W, D

    # -- resident state ------------------------------------------------------------------------------
    def _get_resident(self):
        shared = getattr(self, "_shared", None)
        return shared[0] if shared is not None else self._resident

    def _set_resident(self, state):
        shared = getattr(self, "_shared", None)
        if shared is not None:
            shared[0] = state
        self._resident = state

    @staticmethod
    def _bump(eng):
        """The context is about to change its walkers: DeviceStates handed out before are no longer current."""
        eng.state_epoch = getattr(eng, "state_epoch", 0) + 1

    def _upload_if_needed(self, eng, state, br):
        tc = self.temperature_control
        if isinstance(state, DeviceState) and state.is_current(eng) and (self.trust_resident or not state.materialized):
            return        # the context still holds exactly this state, and nobody has had its arrays in hand to change them
        if self.trust_resident and self._get_resident() is state:
            return
        if state.log_like is None or state.log_prior is None:
            raise ValueError("state must carry log_like and log_prior")
        eng.upload(br.coords[:, :, 0, :], state.log_like, state.log_prior, None if tc is None else tc.betas)
        if tc is not None:
            eng.set_adapt_time(tc.time)

    # -- the tail of every in-model propose(): PT sweep, adaptation, new State -------------------------
    def _finish(self, eng, state, name, br, T, W):
        tc = self.temperature_control
        if self.lazy_state and hasattr(eng, "download_betas"):
            # the walkers stay on the device: the State that goes back reads them when somebody asks (DeviceState)
            if tc is not None and T > 1:
                iperm, i1perm, u = _swap_draws(tc, T, W)
                do_adapt = bool(tc.adaptive)
                sel, swaps = eng.pt_sweep(iperm, i1perm, u, adapt=do_adapt)
                tc.swaps_accepted = swaps
                if do_adapt:
                    tc.betas = eng.download_betas()
                    tc.time += 1
            elif tc is not None:
                tc.swaps_accepted = np.empty(0)
            out = DeviceState(eng, eng.state_epoch, name, (T, W, 1, br.shape[3]), br.inds, betas=None if tc is None else tc.betas,
                              random_state=state.random_state)
            self._set_resident(out)
            return out
        if tc is not None and T > 1:                                   # red_blue.py:330-331, mh.py:190-191
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
        self._set_resident(out)
        return out


    # -- rng="philox": one proposal = one hens_step iteration --------------------------------------------------------------
    def _propose_philox(self, model, state, mh_proposal=None):
        """``propose()`` of the device-draw mode: the context steps ONE iteration of the production path (this move, the swap
        cascade, the ladder adaptation: ensemble.py:974 + red_blue.py:330-331 / mh.py:190-191) and hands back the accept mask,
        the swap counts and the ladder; the walkers stay where they are (DeviceState)."""
        name, br, T, W, D = self._single_branch(state)
        eng = self._ensure_engine(T, W, D)
        if hasattr(eng.likelihood, "evaluate"):
            raise NotImplementedError("rng='philox' steps on the device: it needs a device likelihood (eryn_amd.likelihood)")
        tc = self.temperature_control
        self._apply_periodic(eng, name, D)
        self._upload_if_needed(eng, state, br)
        want = ("mh",) + tuple(np.asarray(v).tobytes() if i else v for i, v in enumerate(mh_proposal)) if mh_proposal else ("stretch",)
        if getattr(eng, "_move_cfg", None) != want:          # which move the context's next iteration runs (weight 1: no mix inside)
            if mh_proposal:
                eng.set_mh_proposal(mh_proposal[0], mh_proposal[1], 1.0)
            else:
                eng.set_mh_proposal(None, None, 0.0)
            eng._move_cfg = want
        self._bump(eng)
        acc, swaps, betas = eng.step_report(1, 1)
        accepted = acc.view(np.bool_)                                  # (one iteration: the counts are 0 / 1)
        if tc is not None:
            tc.swaps_accepted = swaps if T > 1 else np.empty(0)
            if T > 1 and tc.adaptive:
                tc.betas = betas
                tc.time += 1
        if self._accepted is not None:
            # move.accepted (move.py:404-421) is a float array the sampler owns; adding a mask to it costs more host time than the
            # device iteration it describes, so the masks pile up in a byte array and are folded in when somebody looks (the
            # `accepted` property below) or after 200 proposals
            if self._pend is None or self._pend.shape != acc.shape:
                self._fold()
                self._pend, self._npend = np.zeros_like(acc), 0
            np.add(self._pend, acc, out=self._pend)
            self._npend += 1
            if self._npend >= 200:
                self._fold()
        self.num_proposals += 1
        if self.lazy_state:
            out = DeviceState(eng, eng.state_epoch, name, (T, W, 1, D), br.inds, betas=None if tc is None else tc.betas,
                              random_state=state.random_state)
        else:
            x, L, P, _ = eng.download()
            out = State({name: x[:, :, None, :]}, inds={name: br.inds}, log_like=L, log_prior=P,
                        betas=None if tc is None else tc.betas, random_state=state.random_state)
        self._set_resident(out)
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
