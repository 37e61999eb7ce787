"""Periodic parameters of the single branch (eryn/utils/periodic.py:11-47): which parameters are periodic and their periods.

The arithmetic itself (``distance`` / ``wrap``, periodic.py:49-151) runs in the stretch / Metropolis-Hastings kernels
(``hens_set_periodic``); this mirror only carries the information in the reference's form so that a sampler built with
``periodic={branch: {index: period}}`` or with the reference's own container behaves the same.
"""
import numpy as np

__all__ = ["PeriodicContainer", "period_vector"]


class PeriodicContainer:
    def __init__(self, periodic, key_order=None):
        self.periodic = periodic
        self.inds_periodic, self.periods = {}, {}
        for key, spec in periodic.items():
            if spec is None:
                continue
            inds, pers = [], []
            for var, period in spec.items():
                if isinstance(var, str):
                    if key_order is None:
                        raise ValueError("If providing str values for the variable names, must provide key_order argument.")
                    var = key_order[key].index(var)
                inds.append(int(var))
                pers.append(period)
            self.inds_periodic[key] = np.asarray(inds)
            self.periods[key] = np.asarray(pers)


def period_vector(periodic, name, ndim):
    """``[ndim]`` periods for branch ``name`` (0 = not periodic) from a container (this one or the reference's, which
    exposes the same ``inds_periodic`` / ``periods`` dictionaries) or a ``{branch: {index: period}}`` dict; None if the
    branch has no periodic parameter."""
    if periodic is None:
        return None
    if isinstance(periodic, dict):
        periodic = PeriodicContainer(periodic)
    if not (hasattr(periodic, "inds_periodic") and hasattr(periodic, "periods")):
        raise ValueError("periodic must be PeriodicContainer or dict if not None.")      # ensemble.py:340-345
    inds = np.asarray(periodic.inds_periodic.get(name, []), dtype=int)
    pers = np.asarray(periodic.periods.get(name, []), dtype=np.float64)
    if inds.size == 0:
        return None
    if inds.min() < 0 or inds.max() >= ndim:
        raise ValueError(f"periodic parameter index out of range for branch {name!r} with {ndim} parameters")
    if not np.all(pers > 0.0):
        raise ValueError("periods must be positive")
    out = np.zeros(ndim)
    out[inds] = pers
    return out
