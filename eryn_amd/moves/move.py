"""Move base with the attributes EnsembleSampler relies on (eryn/moves/move.py:68-221, 404-457)."""
import numpy as np


class Move:
    def __init__(self, temperature_control=None, periodic=None, gibbs_sampling_setup=None,
                 prevent_swaps=False, skip_supp_names_update=[], is_rj=False, use_gpu=None,
                 random_seed=None, **kwargs):
        if gibbs_sampling_setup is not None:
            raise NotImplementedError("gibbs sampling is outside the device hot path")
        self.periodic = periodic
        self.prevent_swaps = prevent_swaps
        self.is_rj = is_rj
        self.num_proposals = 0
        self.time = 0
        self._accepted = None
        if random_seed is not None:          # move.py:94-96: seeds the *global* stream
            np.random.seed(random_seed)
        self.temperature_control = temperature_control

    # -- periodic parameters (move.py:24-26,82): a PeriodicContainer (this package's or the reference's) or None.  The
    #    reference's sampler also assigns this attribute after construction when it was given one (ensemble.py:528-536).
    #    The device moves hand the periods to their context before every proposal (DeviceMove._apply_periodic).
    @property
    def periodic(self):
        return self._periodic

    @periodic.setter
    def periodic(self, periodic):
        if periodic is not None and not isinstance(periodic, dict) and not (
                hasattr(periodic, "inds_periodic") and hasattr(periodic, "periods")):
            raise ValueError("periodic must be PeriodicContainer or dict if not None.")
        self._periodic = periodic

    # -- counters (move.py:404-421) --------------------------------------------------------------
    @property
    def accepted(self):
        if self._accepted is None:
            raise ValueError("accepted must be inititalized with the init_accepted function if you want to use it.")
        return self._accepted

    @accepted.setter
    def accepted(self, accepted):
        assert isinstance(accepted, np.ndarray)
        self._accepted = accepted

    @property
    def acceptance_fraction(self):
        return self.accepted / self.num_proposals

    # -- tempering wiring (move.py:428-441) --------------------------------------------------------
    @property
    def temperature_control(self):
        return self._temperature_control

    @temperature_control.setter
    def temperature_control(self, temperature_control):
        self._temperature_control = temperature_control
        self.ntemps = 1 if temperature_control is None else temperature_control.ntemps

    def compute_log_posterior_basic(self, logl, logp):
        return logl + logp

    def tune(self, state, accepted):
        """Place holder for tuning, called by the sampler after every proposal when ``tune=True`` (move.py:459-470,
        ensemble.py:983-984)."""
        pass

    def propose(self, model, state):
        raise NotImplementedError("The proposal must be implemented by subclasses")
