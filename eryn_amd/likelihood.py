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

    def _install(self, lib, ctx, row_width=None):
        """``row_width`` > ndim: the context's rows are padded (engine.HipEnsemble): zero mean and zero rows / columns of the
        precision matrix on the pads, which therefore never enter the quadratic form."""
        mu, prec = self.mu, self.invcov
        if row_width is not None and row_width != self.ndim:
            D, R = self.ndim, int(row_width)
            mu = np.zeros(R)
            mu[:D] = self.mu
            prec = np.zeros((R, R) if self.invcov.ndim == 2 else R)
            if self.invcov.ndim == 2:
                prec[:D, :D] = self.invcov
            else:
                prec[:D] = self.invcov
        self._installed = (np.ascontiguousarray(mu), np.ascontiguousarray(prec))     # (alive for the duration of the call)
        _lib.check(lib.hens_set_gaussian(ctx, _lib.ptr(self._installed[0]), _lib.ptr(self._installed[1])), ctx)


class RosenbrockLikelihood:
    """``-(sum_i b (x[i+1] - x[i]^2)^2 + (a - x[i])^2)`` (BASELINE config 5 stress target)."""

    kind = _lib.LIKE_ROSENBROCK

    def __init__(self, ndim, a=1.0, b=100.0):
        self.ndim, self.a, self.b = int(ndim), float(a), float(b)

    def _install(self, lib, ctx):
        _lib.check(lib.hens_set_rosenbrock(ctx, self.a, self.b), ctx)


class HostLikelihood:
    """An arbitrary Python ``log_like_fn`` (the reference's ``_FunctionWrapper``, ensemble.py:1623-1667).

    The proposal, prior, accept test and update still run on the MI355X; only the likelihood is
    evaluated here, on the proposed points that lie inside the prior support, between
    ``hens_propose_split`` and ``hens_accept_split``.  This is the API-complete slow path (two PCIe hops
    per half-step); the built-in likelihoods above are the fused fast path.

    ``vectorize=True``: ``f(x[N, ndim], *args, **kwargs) -> [N]``; otherwise one call per walker with a
    1-D ``x`` (ensemble.py:1371-1481).
    """

    kind = _lib.LIKE_HOST

    def __init__(self, f, ndim, args=None, kwargs=None, vectorize=True, fill_value=-1e300):
        self.f, self.ndim = f, int(ndim)
        self.args = [] if args is None else list(args)
        self.kwargs = {} if kwargs is None else dict(kwargs)
        self.vectorize, self.fill_value = bool(vectorize), float(fill_value)

    def _install(self, lib, ctx):
        pass

    def evaluate(self, q, inbox):
        """q[T, N, D], inbox[T, N] -> logl[T, N] with the reference's contract (ensemble.py:1219-1545):
        walkers outside the prior support are not evaluated and get the fill value."""
        import warnings
        T, N, D = q.shape
        if np.any(np.isinf(q)):
            raise ValueError("At least one parameter value was infinite")
        if np.any(np.isnan(q)):
            raise ValueError("At least one parameter value was NaN")
        valid = np.asarray(inbox, dtype=bool).reshape(-1)
        ll = np.full(T * N, -1e300)
        if not valid.any():
            warnings.warn("All points input for the Likelihood have a log prior of -inf.")
            return ll.reshape(T, N)
        x = q.reshape(-1, D)[valid]
        if self.vectorize:
            res = np.asarray(self.f(x, *self.args, **self.kwargs), dtype=np.float64)
        else:
            res = np.asarray([self.f(xi, *self.args, **self.kwargs) for xi in x], dtype=np.float64)
        if res.ndim == 2 and res.shape[1] == 1:
            res = np.squeeze(res, axis=1)
        if res.shape != (x.shape[0],):
            raise ValueError("log_like_fn must return one value per walker (blobs are outside the device path)")
        ll[valid] = res
        ll[~valid] = self.fill_value
        if np.any(np.isnan(ll)):
            raise ValueError("The likelihood function is returning Nan.")
        return ll.reshape(T, N)
