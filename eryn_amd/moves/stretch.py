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

from .device import DeviceMove

__all__ = ["StretchMove"]


class StretchMove(DeviceMove):
    """Goodman-Weare stretch move evaluated on the MI355X.

    Args beyond the reference's ``StretchMove(a=2.0, nsplits=2, randomize_split=True,
    live_dangerously=False)``:
        likelihood: a :mod:`eryn_amd.likelihood` object (the likelihood lives inside the kernel).
        prior_box: ``(lo, hi)`` scalars or ``[ndim]`` arrays, or a ``ProbDistContainer``.
        device_id, fill_value: engine options.
        trust_resident: skip the host->device upload when ``state`` is the object this move
            returned last (safe when no host-side code mutates the State in between).
        rng: "numpy" (default) - the reference's streams drawn on the host in the reference's order: the reference's chain;
            "philox" - ``propose()`` is ONE ``hens_step`` iteration with device-side draws (the benchmarked kernels).
        lazy_state: return a ``DeviceState`` whose arrays are copied from the device when they are read (default with
            rng="philox"): a sampler that stores every ``thin_by``-th step downloads the walkers at stored steps only.
        seed: Philox seed of a context this move creates itself (default: drawn from the global ``np.random``).
    """

    needs_walker_guard = True          # red_blue.py:108-114

    def __init__(self, a=2.0, nsplits=2, randomize_split=True, live_dangerously=False, likelihood=None,
                 prior_box=None, device_id=0, fill_value=-1e300, trust_resident=False, return_gpu=False,
                 rng="numpy", lazy_state=None, seed=None, **kwargs):
        if not 2 <= int(nsplits) <= 8:
            raise NotImplementedError("the device path runs red-blue moves of 2 to 8 sets")
        self.a = a
        self.nsplits = nsplits
        self.randomize_split = randomize_split
        self.live_dangerously = live_dangerously
        DeviceMove.__init__(self, likelihood=likelihood, prior_box=prior_box, device_id=device_id,
                            fill_value=fill_value, trust_resident=trust_resident, a=a,
                            live_dangerously=live_dangerously, rng=rng, lazy_state=lazy_state, seed=seed, **kwargs)

    # -- the plugin entry point -------------------------------------------------------------------
    def propose(self, model, state):
        name, br, T, W, D = self._single_branch(state)
        eng = self._ensure_engine(T, W, D)
        if getattr(eng, "a", self.a) != float(self.a):                  # (a tuning hook changed move.a: stretch.py:37 reads it per proposal)
            eng.set_stretch_scale(self.a)
        if getattr(eng, "nsplits", 2) != self.nsplits:                 # (the context's parity API defaults to two sets)
            eng.set_nsplits(self.nsplits)
        if self.rng == "philox":
            return self._propose_philox(model, state)
        self._apply_periodic(eng, name, D)
        self._upload_if_needed(eng, state, br)
        self._bump(eng)

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
        return self._finish(eng, state, name, br, T, W), accepted
