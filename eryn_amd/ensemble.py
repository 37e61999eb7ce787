"""``EnsembleSampler`` with the reference's constructor / ``sample`` / ``run_mcmc`` contract
(eryn/ensemble.py:211-681, 808-1125) for the stretch + parallel-tempering path, driving the
MI355X engine.

Two RNG modes:
  rng="numpy"   the reference's streams (sampler-owned RandomState cloned from the global
                ``np.random`` at construction + the global stream itself) drive the device
                kernels -> same chain as Eryn for the same seeds (drop-in / parity mode).
  rng="philox"  device-side counter-based Philox, ``thin_by`` iterations per host call with no
                host round trip (production mode; statistically equivalent sampler).
"""
import numpy as np

from .backend import Backend
from .engine import HipEnsemble
from .model import Model
from .moves import DeviceMove, GaussianMove, MHMove, StretchMove, TemperatureControl
from .periodic import PeriodicContainer, period_vector
from .prior import ProbDistContainer
from .state import State


class EnsembleSampler:
    def __init__(self, nwalkers, ndims, log_like_fn, priors, tempering_kwargs={}, branch_names=None,
                 nleaves_max=1, moves=None, args=None, kwargs=None, backend=None, vectorize=True, periodic=None,
                 fill_zero_leaves_val=-1e300, rng="numpy", seed=None, device_id=0, info={}, num_repeats_in_model=1,
                 num_repeats_rj=1, **unused):
        # -- shapes: a single branch with one leaf (ensemble.py:265-317 normalises to dicts)
        if isinstance(ndims, dict):
            if len(ndims) != 1:
                raise NotImplementedError("the device path handles a single branch")
            branch_names = list(ndims.keys())
            ndims = list(ndims.values())[0]
        if isinstance(ndims, (list, tuple, np.ndarray)):
            if len(ndims) != 1:
                raise NotImplementedError("the device path handles a single branch")
            ndims = int(ndims[0])
        if isinstance(nleaves_max, dict):
            nleaves_max = list(nleaves_max.values())[0]
        if isinstance(nleaves_max, (list, tuple, np.ndarray)):
            nleaves_max = int(nleaves_max[0])
        if nleaves_max != 1:
            raise NotImplementedError("the device path handles nleaves_max == 1 (RJ is a later row, SURVEY 8f-4)")
        if not hasattr(log_like_fn, "_install"):
            # an arbitrary Python callable (the reference's contract): proposal / prior / accept / update /
            # PT on the device, the likelihood here on the host, two PCIe hops per half-step
            if not callable(log_like_fn):
                raise ValueError("log_like_fn must be callable or an eryn_amd.likelihood object")
            if rng != "numpy":
                raise NotImplementedError("a host-callable likelihood needs rng='numpy' (the device cannot call Python)")
            from .likelihood import HostLikelihood
            log_like_fn = HostLikelihood(log_like_fn, int(ndims), args=args, kwargs=kwargs, vectorize=vectorize,
                                         fill_value=fill_zero_leaves_val)
        self.nwalkers, self.ndim = int(nwalkers), int(ndims)
        # ensemble.py:243-256: the in-model step runs num_repeats_in_model times per sampler iteration (num_repeats_rj belongs to
        # the between-model step, which a single fixed-dimension branch does not have: kept for the constructor contract)
        self.num_repeats_in_model, self.num_repeats_rj = int(num_repeats_in_model), int(num_repeats_rj)
        if self.num_repeats_in_model < 1 or self.num_repeats_rj < 1:
            raise ValueError("num_repeats_in_model / num_repeats_rj must be >= 1")
        self.branch_names = ["model_0"] if branch_names is None else list(branch_names)
        self.ndims = {self.branch_names[0]: self.ndim}
        self.nleaves_max = {self.branch_names[0]: 1}
        self.log_like_fn = log_like_fn
        self.fill_zero_leaves_val = fill_zero_leaves_val
        self.rng = rng
        if rng not in ("numpy", "philox"):
            raise ValueError("rng must be 'numpy' or 'philox'")

        # -- priors (ensemble.py:334-348): dict of dists, a container, or {branch: container}
        if isinstance(priors, dict) and self.branch_names[0] in priors:
            priors = priors[self.branch_names[0]]
        if isinstance(priors, dict):
            priors = ProbDistContainer(priors)
        if not hasattr(priors, "box_bounds"):
            raise NotImplementedError("priors must be uniform box priors (eryn_amd.prior.ProbDistContainer)")
        self.priors = {self.branch_names[0]: priors}
        lo, hi = priors.box_bounds()

        # -- tempering (ensemble.py:321-332): enabled iff tempering_kwargs != {}
        if tempering_kwargs == {}:
            self.temperature_control, self.ntemps = None, 1
        else:
            self.temperature_control = TemperatureControl(self.ndim, self.nwalkers, **tempering_kwargs)
            self.ntemps = self.temperature_control.ntemps
        tc = self.temperature_control

        # -- periodic parameters (ensemble.py:338-347): a dict becomes a container; moves without one of their own get it
        if periodic is not None:
            if isinstance(periodic, dict):
                periodic = PeriodicContainer(periodic)
            elif not (hasattr(periodic, "inds_periodic") and hasattr(periodic, "periods")):
                raise ValueError("periodic must be PeriodicContainer or dict if not None.")
        self.periodic = periodic

        # -- moves (ensemble.py:350-378, 517-544)
        if moves is None:
            moves = [StretchMove(a=2.0, periodic=periodic)]
        elif not isinstance(moves, (list, tuple)):
            moves = [moves]
        self.moves, weights = [], []
        for m in moves:
            m, w = m if isinstance(m, (list, tuple)) else (m, 1.0)
            self.moves.append(m)
            weights.append(w)
        self.weights = np.atleast_1d(weights).astype(float)
        self.weights /= np.sum(self.weights)

        kw = {}
        if tc is not None:
            kw = dict(adaptive=tc.adaptive, adaptation_lag=tc.adaptation_lag, adaptation_time=tc.adaptation_time,
                      stop_adaptation=tc.stop_adaptation)
        for m in self.moves:
            if not isinstance(m, DeviceMove):
                raise NotImplementedError("only eryn_amd.moves.StretchMove / GaussianMove run on the device path")
        stretch_moves = [m for m in self.moves if isinstance(m, StretchMove)]
        # the W >= 2 ndim guard belongs to the red-blue move (red_blue.py:108-114): without one it does not apply
        live = any(m.live_dangerously for m in stretch_moves) or not stretch_moves
        a_vals = {m.a for m in stretch_moves} or {2.0}
        if len(a_vals) != 1:
            raise NotImplementedError("all device stretch moves must share one scale a")
        if seed is None:
            seed = int(np.random.randint(0, 2**31 - 1)) if rng == "philox" else 0
        self.seed = int(seed)
        self.engine = HipEnsemble(self.ntemps, self.nwalkers, self.ndim, log_like_fn, lo, hi, a=a_vals.pop(),
                                  tempered=tc is not None, live_dangerously=live,
                                  fill_value=fill_zero_leaves_val, seed=seed, device_id=device_id, **kw)
        self._resident = [None]              # the State the device context mirrors, shared by the moves
        for m in self.moves:
            if m.temperature_control is None:
                m.temperature_control = tc
            if periodic is not None and m.periodic is None:          # ensemble.py:528-536
                m.periodic = periodic
            m.attach_engine(self.engine, self._resident)
            m.trust_resident = True
            m.accepted = np.zeros((self.ntemps, self.nwalkers))
        # device-side draws (rng="philox"): at most one stretch move and one Gaussian MH move, mixed by weight
        self._philox_moves = None
        if rng == "philox":
            mh = [(m, w) for m, w in zip(self.moves, self.weights) if isinstance(m, MHMove)]
            st = [(m, w) for m, w in zip(self.moves, self.weights) if isinstance(m, StretchMove)]
            if len(mh) > 1 or len(st) > 1 or any(not isinstance(m, GaussianMove) for m, _ in mh):
                raise NotImplementedError("rng='philox' mixes at most one StretchMove with one GaussianMove")
            self._mh_weight = float(mh[0][1]) if mh else 0.0
            self._mh_pushed = None
            if mh:
                self._push_mh_proposal(mh[0][0])
            self._philox_moves = (st[0][0] if st else None, mh[0][0] if mh else None)

        # -- backend + RNG (ensemble.py:593-652)
        self.backend = Backend() if backend is None else backend
        if not self.backend.initialized:
            self.backend.reset(self.nwalkers, self.ndims, ntemps=self.ntemps, branch_names=self.branch_names)
        self._random = np.random.mtrand.RandomState()
        self._random.set_state(np.random.get_state())       # R := snapshot of the global stream
        self._previous_state = None

    # -- reference accessors ----------------------------------------------------------------------
    @property
    def random_state(self):
        return self._random.get_state()

    @random_state.setter
    def random_state(self, state):
        try:
            self._random.set_state(state)
        except Exception:
            pass

    @property
    def iteration(self):
        return self.backend.iteration

    @property
    def acceptance_fraction(self):
        return self.backend.accepted / float(max(self.backend.iteration, 1))

    def get_model(self):
        """ensemble.py:780-806."""
        return Model(self.log_like_fn, self.compute_log_like, self.compute_log_prior,
                     self.temperature_control, map, self._random)

    def _eval(self, coords):
        x = coords[self.branch_names[0]] if isinstance(coords, dict) else coords
        x = np.asarray(x)
        if x.ndim == 4:
            x = x[:, :, 0, :]
        tc = self.temperature_control
        self.engine.upload(x, betas=None if tc is None else tc.betas)
        self.engine.eval_state()
        _, L, P, _ = self.engine.download(want_x=False)
        if hasattr(self.log_like_fn, "evaluate"):            # host-callable likelihood: the device filled log_prior only
            L = self.log_like_fn.evaluate(np.ascontiguousarray(x), ~np.isinf(P))
        self._resident[0] = None
        for m in self.moves:
            m._resident = None
        return L, P

    def compute_log_prior(self, coords, inds=None, supps=None, branch_supps=None):
        """[ntemps, nwalkers] log-prior, evaluated on the device (ensemble.py:1127-1217)."""
        return self._eval(coords)[1]

    def compute_log_like(self, coords, inds=None, logp=None, supps=None, branch_supps=None):
        """([ntemps, nwalkers] log-like, blobs=None), evaluated on the device (ensemble.py:1219-1545):
        walkers outside the prior support are not evaluated and get ``fill_zero_leaves_val``."""
        return self._eval(coords)[0], None

    # -- main loop (ensemble.py:808-1045) ------------------------------------------------------------
    def sample(self, initial_state, iterations=1, tune=False, skip_initial_state_check=True, thin_by=1,
               store=True, progress=False):
        state = State(initial_state, copy=True)
        name = self.branch_names[0]
        if state.branches[name].shape != (self.ntemps, self.nwalkers, 1, self.ndim):
            raise ValueError("incompatible input dimensions")
        if state.log_prior is None or state.log_like is None:
            L, P = self._eval(state.branches_coords)
            state.log_prior = P if state.log_prior is None else state.log_prior
            state.log_like = L if state.log_like is None else state.log_like
        tc = self.temperature_control
        if tc is not None:
            if state.betas is not None:
                if state.betas.shape[0] != self.ntemps:
                    raise ValueError("Input state has inverse temperatures (betas), but not the correct number.")
                tc.betas = state.betas.copy()
            else:
                state.betas = tc.betas.copy()
        if np.any(np.isinf(state.log_like)):
            raise ValueError("The initial log_like was +/- infinite")
        if np.any(np.isinf(state.log_prior)):
            raise ValueError("The initial log_prior was +/- infinite")
        if np.any(np.isnan(state.log_like)) or np.any(np.isnan(state.log_prior)):
            raise ValueError("The initial log_like / log_prior was NaN")
        thin_by = int(thin_by)
        if thin_by <= 0:
            raise ValueError("Invalid thinning argument")
        if store:
            self.backend.grow(iterations, None)
        model = self.get_model()

        if self.rng == "philox":
            yield from self._sample_philox(state, iterations, thin_by, store, tune)
            return

        for _ in range(iterations):
            for sub in range(thin_by):
                # the reference re-zeroes `accepted` at every sub-iteration (ensemble.py:968): what reaches the backend is the LAST
                # sub-iteration's sum over its repeats - only that one is formed here; and `state.random_state` (ensemble.py:981,
                # a copy of R's 2.5 kB state: ~50 us) is attached where somebody can see the state: the tune hook, the yield
                last = sub == thin_by - 1
                if last:
                    accepted = np.zeros((self.ntemps, self.nwalkers))
                for _repeat in range(self.num_repeats_in_model):                # ensemble.py:969-984
                    move = self._random.choice(self.moves, p=self.weights)      # ensemble.py:971
                    state, accepted_out = move.propose(model, state)
                    if last:
                        accepted += accepted_out
                    if tune:
                        state.random_state = self.random_state
                        move.tune(state, accepted_out)
            swaps = tc.swaps_accepted if self.ntemps > 1 else None
            state.random_state = self.random_state
            if store:
                self.backend.save_step(state, accepted, swaps_accepted=swaps)
            self._previous_state = state
            yield state

    def _push_mh_proposal(self, mh_move):
        """Hand the Gaussian move's proposal to the device if it differs from what the device holds (construction; a ``tune`` hook
        that rescales the move, ensemble.py:983-984 - the reference's hook mutates the move object the next proposal reads)."""
        kind, scale = mh_move.device_proposal()
        key = (kind, np.asarray(scale, dtype=np.float64).tobytes())
        if key != self._mh_pushed:
            self.engine.set_mh_proposal(kind, scale, self._mh_weight)
            self._mh_pushed = key

    def philox_checkpoint(self):
        """What a stored State carries as ``random_state`` in Philox mode: the device draws are a pure function of (seed,
        iteration, rung, walker), so (seed, iteration counter, adaptation time) is the whole generator state - the device-side
        form of the reference's per-step checkpoint of R (backends/backend.py:1014-1091, ensemble.py:605-647)."""
        tc = self.temperature_control
        return ("philox", self.seed, int(self.engine.iteration()), 0 if tc is None else int(tc.time))

    def _sample_philox(self, state, iterations, thin_by, store, tune=False):
        eng, tc, name = self.engine, self.temperature_control, self.branch_names[0]
        eng.upload(state.branches[name].coords[:, :, 0, :], state.log_like, state.log_prior,
                   None if tc is None else tc.betas)
        rs = state.random_state
        if isinstance(rs, tuple) and len(rs) == 4 and rs[0] == "philox":      # resume a stored chain on its own stream
            if int(rs[1]) != self.seed:
                raise ValueError(f"the state was stored by a sampler with seed {rs[1]}, this one has seed {self.seed}")
            eng.set_iteration(int(rs[2]))
            if tc is not None:
                tc.time = int(rs[3])
        if tc is not None:
            eng.set_adapt_time(tc.time)
        reps = self.num_repeats_in_model
        st_move, mh_move = self._philox_moves
        # one context steps both moves of the mix: they must agree on the periodic parameters
        pers = [period_vector(m.periodic, name, self.ndim) for m in (st_move, mh_move) if m is not None]
        if len(pers) == 2 and not ((pers[0] is None and pers[1] is None) or
                                   (pers[0] is not None and pers[1] is not None and np.array_equal(pers[0], pers[1]))):
            raise NotImplementedError("rng='philox': the moves of a mix must share their periodic parameters")
        eng.set_periodic(pers[0] if pers else None)
        if st_move is not None and getattr(eng, "nsplits", 2) != st_move.nsplits:      # RedBlueMove(nsplits=...), red_blue.py:41-47
            eng.set_nsplits(st_move.nsplits)
        if st_move is not None and eng.a != float(st_move.a):
            eng.set_stretch_scale(st_move.a)
        prev = eng.counters()
        prev_mh = eng.mh_counters() if mh_move is not None else None
        inds = state.branches[name].inds
        for _ in range(iterations):
            # the reference stores the LAST thinned step's accept mask and swap counts (ensemble.py:968-979,
            # 1013-1024: `accepted` is re-zeroed every sub-iteration); the moves' own counters see every step
            # (num_repeats_in_model device iterations make one sampler sub-iteration: its accept mask sums over the repeats,
            #  ensemble.py:968-975)
            # (thin_by > 1: ONE device call; the counters in front of the last sub-iteration stay on the device until the
            #  download below - no split call, no counter read in between)
            mid_acc, mid_mh_acc = prev["accepted"], None if prev_mh is None else prev_mh["accepted"]
            if tune:
                # the reference calls move.tune(state, accepted_out) after EVERY proposal with that proposal's own mask
                # (ensemble.py:969-984): a host hook per proposal, so the device steps one iteration at a time here and the
                # move that ran is read off the counters
                last, last_mh = prev, prev_mh
                for _sub in range(thin_by):
                    if _sub == thin_by - 1:
                        mid_acc = last["accepted"]
                        mid_mh_acc = None if last_mh is None else last_mh["accepted"]
                    for _rep in range(reps):
                        eng.step(1)
                        c1 = eng.counters()
                        cm1 = eng.mh_counters() if mh_move is not None else None
                        ran_stretch = c1["num_proposals"] > last["num_proposals"]
                        out = c1["accepted"] - last["accepted"] if ran_stretch else cm1["accepted"] - last_mh["accepted"]
                        xi, Li, Pi, bi = eng.download()
                        if tc is not None:
                            tc.time = c1["adapt_time"]             # (the checkpoint in the State the hook sees: this proposal's)
                        st_i = State({name: xi[:, :, None, :]}, inds={name: inds}, log_like=Li, log_prior=Pi,
                                     betas=None if tc is None else bi, random_state=self.philox_checkpoint())
                        (st_move if ran_stretch else mh_move).tune(st_i, out)
                        if st_move is not None and eng.a != float(st_move.a):      # (the hook may retune the stretch scale)
                            eng.set_stretch_scale(st_move.a)
                        if mh_move is not None:                                    # (... or the Gaussian move's scale / covariance)
                            self._push_mh_proposal(mh_move)
                        last, last_mh = c1, cm1
            elif thin_by > 1 and hasattr(eng, "step_marked"):
                eng.step_marked((thin_by - 1) * reps, reps)
            else:
                if thin_by > 1:
                    eng.step((thin_by - 1) * reps)
                    mid_acc = eng.counters()["accepted"]
                    mid_mh_acc = eng.mh_counters()["accepted"] if mh_move is not None else None
                eng.step(reps)
            x, L, P, betas = eng.download()
            c = eng.counters()
            if not tune and thin_by > 1 and hasattr(eng, "step_marked"):
                mid_acc, mid_mh_acc = eng.marked_counters()
            accepted = c["accepted"] - mid_acc
            if st_move is not None:
                st_move.accepted += c["accepted"] - prev["accepted"]
                st_move.num_proposals += c["num_proposals"] - prev["num_proposals"]
            if mh_move is not None:
                cm = eng.mh_counters()
                mh_move.accepted += cm["accepted"] - prev_mh["accepted"]
                mh_move.num_proposals += cm["num_proposals"] - prev_mh["num_proposals"]
                accepted = accepted + (cm["accepted"] - mid_mh_acc)
                prev_mh = cm
            swaps = None
            if tc is not None:
                tc.betas = betas
                tc.time = c["adapt_time"]
                tc.swaps_accepted = c["swaps_last"]
                swaps = c["swaps_last"]
            prev = c
            state = State({name: x[:, :, None, :]}, inds={name: inds}, log_like=L, log_prior=P,
                          betas=None if tc is None else betas, random_state=self.philox_checkpoint())
            if store:
                self.backend.save_step(state, accepted, swaps_accepted=swaps)
            self._previous_state = state
            yield state

    def run_mcmc(self, initial_state, nsteps, burn=None, post_burn_update=False, **kwargs):
        """ensemble.py:1047-1125."""
        if initial_state is None:
            if self._previous_state is None:
                raise ValueError("Cannot have `initial_state=None` if run_mcmc has never been called.")
            initial_state = self._previous_state
        if burn is not None and burn != 0:
            bk = dict(kwargs)
            bk.update(store=False, thin_by=1)
            for initial_state in self.sample(initial_state, iterations=burn, **bk):
                pass
        if nsteps == 0:                                    # ensemble.py:1095-1096
            return initial_state
        results = None
        if kwargs.get("store", True) is False:
            # nothing is stored: ONE yield (rng="philox": one device-resident call for all iterations, one download at the end;
            # rng="numpy": the same proposals in the same order, the state is looked at once, at the end)
            kw = dict(kwargs)
            thin = int(kw.pop("thin_by", 1))
            for results in self.sample(initial_state, iterations=1, thin_by=thin * nsteps, **kw):
                pass
        else:
            for results in self.sample(initial_state, iterations=nsteps, **kwargs):
                pass
        self._previous_state = results
        return results

    # -- chain accessors --------------------------------------------------------------------------
    def get_chain(self, **kw):
        return self.backend.get_chain(**kw)

    def get_log_like(self, **kw):
        return self.backend.get_log_like(**kw)

    def get_log_prior(self, **kw):
        return self.backend.get_log_prior(**kw)

    def get_betas(self, **kw):
        return self.backend.get_betas(**kw)
