"""Device-resident likelihood kinds of the HIP stepping engine.

The reference takes an arbitrary Python ``log_like_fn`` (ensemble.py:1623-1667).  The hot
path evaluated on the MI355X needs the likelihood inside the fused kernel, so the
likelihoods of the BASELINE configs are built in and selected by passing one of these
objects as ``log_like_fn``.
"""
import numpy as np

from . import _lib


class GaussianLikelihood:
    """``-0.5 (x - mu)^T invcov (x - mu)``: the reference tests' ``log_like_fn(x, mu, invcov)``
    (tests/test_eryn.py:33-35).  ``invcov`` is ``[D, D]`` (dense) or ``[D]`` (diagonal)."""

    def __init__(self, mu, invcov):
        self.mu = np.ascontiguousarray(mu, dtype=np.float64)
        self.invcov = np.ascontiguousarray(invcov, dtype=np.float64)
        D = self.mu.shape[0]
        if self.invcov.shape == (D, D):
            self.kind = _lib.LIKE_GAUSS_DENSE
        elif self.invcov.shape == (D,):
            self.kind = _lib.LIKE_GAUSS_DIAG
        else:
            raise ValueError("invcov must have shape (D, D) or (D,)")
        self.ndim = D

    def _install(self, lib, ctx):
        _lib.check(lib.hens_set_gaussian(ctx, _lib.ptr(self.mu), _lib.ptr(self.invcov)), ctx)


class RosenbrockLikelihood:
    """``-(sum_i b (x[i+1] - x[i]^2)^2 + (a - x[i])^2)`` (BASELINE config 5 stress target)."""

    kind = _lib.LIKE_ROSENBROCK

    def __init__(self, ndim, a=1.0, b=100.0):
        self.ndim, self.a, self.b = int(ndim), float(a), float(b)

    def _install(self, lib, ctx):
        _lib.check(lib.hens_set_rosenbrock(ctx, self.a, self.b), ctx)
