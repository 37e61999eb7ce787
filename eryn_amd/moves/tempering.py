"""Temperature ladder control with the reference's surface (eryn/moves/tempering.py).

``make_ladder`` builds the default ladder on the host (init-time only).  The swap cascade
and the ladder adaptation run on the MI355X (``hens_pt_sweep``): ``TemperatureControl``
here only owns the ladder, draws the reference's random numbers in the reference's
order when driven in NumPy-RNG mode, and mirrors the counters back.
"""
import numpy as np

__all__ = ["TemperatureControl", "make_ladder"]

# 25 %-swap temperature steps for a D-dimensional Gaussian, D = 1..100
# (tempering.py:57-160; tabulated by ptemcee, Vousden et al. 2016).
_STEP_25 = (
    25.2741, 7., 4.47502, 3.5236, 3.0232, 2.71225, 2.49879, 2.34226, 2.22198, 2.12628,
    2.04807, 1.98276, 1.92728, 1.87946, 1.83774, 1.80096, 1.76826, 1.73895, 1.7125, 1.68849,
    1.66657, 1.64647, 1.62795, 1.61083, 1.59494, 1.58014, 1.56632, 1.55338, 1.54123, 1.5298,
    1.51901, 1.50881, 1.49916, 1.49, 1.4813, 1.47302, 1.46512, 1.45759, 1.45039, 1.4435,
    1.4369, 1.43056, 1.42448, 1.41864, 1.41302, 1.40761, 1.40239, 1.39736, 1.3925, 1.38781,
    1.38327, 1.37888, 1.37463, 1.37051, 1.36652, 1.36265, 1.35889, 1.35524, 1.3517, 1.34825,
    1.3449, 1.34164, 1.33847, 1.33538, 1.33236, 1.32943, 1.32656, 1.32377, 1.32104, 1.31838,
    1.31578, 1.31325, 1.31076, 1.30834, 1.30596, 1.30364, 1.30137, 1.29915, 1.29697, 1.29484,
    1.29275, 1.29071, 1.2887, 1.28673, 1.2848, 1.28291, 1.28106, 1.27923, 1.27745, 1.27569,
    1.27397, 1.27227, 1.27061, 1.26898, 1.26737, 1.26579, 1.26424, 1.26271, 1.26121, 1.25973)


def make_ladder(ndim, ntemps=None, Tmax=None):
    """Geometric ladder of inverse temperatures (tempering.py:10-197), same argument rules."""
    if type(ndim) != int or ndim < 1:
        raise ValueError("Invalid number of dimensions specified.")
    if ntemps is None and Tmax is None:
        raise ValueError("Must specify one of ``ntemps`` and ``Tmax``.")
    if Tmax is not None and Tmax <= 1:
        raise ValueError("``Tmax`` must be greater than 1.")
    if ntemps is not None and (type(ntemps) != int or ntemps < 1):
        raise ValueError("Invalid number of temperatures specified.")
    table = np.array(_STEP_25)
    step = table[ndim - 1] if ndim <= table.shape[0] else 1.0 + 2.0 * np.sqrt(np.log(4.0)) / np.sqrt(ndim)
    hot_rung_at_infinity = Tmax == np.inf
    if hot_rung_at_infinity:
        Tmax, ntemps = None, ntemps - 1
    if ntemps is None:
        if Tmax is None:
            raise ValueError("Must specify at least one of ``ntemps`` and finite ``Tmax``.")
        ntemps = int(np.log(Tmax) / np.log(step) + 2)
    elif Tmax is None:
        Tmax = step ** (ntemps - 1)
    betas = np.logspace(0, -np.log10(Tmax), ntemps)
    return np.concatenate((betas, [0])) if hot_rung_at_infinity else betas


class TemperatureControl:
    """Ladder owner (tempering.py:200-282).  Same constructor arguments and attributes."""

    def __init__(self, effective_ndim, nwalkers, ntemps=1, betas=None, Tmax=None, adaptive=True,
                 adaptation_lag=10000, adaptation_time=100, stop_adaptation=-1, permute=True,
                 skip_swap_supp_names=[]):
        if betas is None:
            betas = np.array([1.0]) if ntemps == 1 else make_ladder(effective_ndim, ntemps=ntemps, Tmax=Tmax)
        self.nwalkers = nwalkers
        self.betas = np.array(betas, dtype=np.float64)
        self.ntemps = len(self.betas)
        self.permute = permute
        self.skip_swap_supp_names = skip_swap_supp_names
        self.time = 0
        self.adaptive = adaptive
        self.adaptation_time, self.adaptation_lag = adaptation_time, adaptation_lag
        self.stop_adaptation = stop_adaptation
        self.swaps_proposed = np.full(self.ntemps - 1, self.nwalkers)
        self.swaps_accepted = np.zeros(self.ntemps - 1)

    def compute_log_posterior_tempered(self, logl, logp, betas=None):
        """Host helper with the reference's semantics (tempering.py:284-349); the device kernel
        applies the same rule inside the accept test."""
        assert logl.shape == logp.shape
        return self.tempered_likelihood(logl, betas=betas) + logp

    def tempered_likelihood(self, logl, betas=None):
        if logl.ndim == 1:
            if betas is None:
                raise ValueError("If inputing a 1D logl array, need to provide 1D betas array of the same length.")
            out = logl * betas
        else:
            betas = self.betas if betas is None else betas
            with np.errstate(invalid="ignore"):
                out = logl * betas[:, None]
        out[np.isnan(out)] = -np.inf
        return out

    def draw_swap_randoms(self):
        """The cascade's draws from the process-global ``np.random`` stream, in the reference's
        order: per pair (hot -> cold) permutation, permutation, uniform (tempering.py:515-535)."""
        T, W = self.ntemps, self.nwalkers
        iperm = np.empty((T - 1, W), dtype=np.int64)
        i1perm = np.empty((T - 1, W), dtype=np.int64)
        u = np.empty((T - 1, W))
        for j in range(T - 1):
            if self.permute:
                iperm[j] = np.random.permutation(W)
                i1perm[j] = np.random.permutation(W)
            else:
                iperm[j] = np.arange(W)
                i1perm[j] = np.arange(W)
            u[j] = np.random.uniform(size=W)
        return iperm, i1perm, u
