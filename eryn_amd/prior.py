"""Box priors with the reference's names (eryn/prior.py:12-136, 219-497), reduced to what the
device hot path supports: independent uniform distributions per parameter.  The engine
evaluates the prior inside the fused kernel; these objects only describe the box and draw
starting points."""
import numpy as np


class UniformDistribution:
    def __init__(self, min_val, max_val):
        if min_val > max_val:
            min_val, max_val = max_val, min_val
        elif min_val == max_val:
            raise ValueError("Min and max values are the same.")
        self.min_val, self.max_val = min_val, max_val
        self.diff = max_val - min_val
        self.pdf_val = 1 / self.diff
        self.logpdf_val = np.log(self.pdf_val)

    def rvs(self, size=1):
        if not isinstance(size, (int, tuple)):
            raise ValueError("size must be an integer or tuple of ints.")
        if isinstance(size, int):
            size = (size,)
        return np.random.rand(*size) * self.diff + self.min_val


def uniform_dist(min, max):
    return UniformDistribution(min, max)


class ProbDistContainer:
    """``{index: distribution}`` container (prior.py:219-335).  Only single-index uniform entries
    are accepted: anything else cannot run inside the kernel."""

    def __init__(self, priors_in):
        self.priors_in = dict(priors_in)
        keys = sorted(self.priors_in)
        if keys != list(range(len(keys))):
            raise ValueError("prior keys must be the integers 0..ndim-1")
        for k, dist in self.priors_in.items():
            if not isinstance(dist, UniformDistribution):
                raise NotImplementedError("the device path supports uniform (box) priors only")
        self.ndim = len(keys)
        self.priors = [([k], self.priors_in[k]) for k in keys]

    def box_bounds(self):
        lo = np.array([self.priors_in[k].min_val for k in range(self.ndim)], dtype=np.float64)
        hi = np.array([self.priors_in[k].max_val for k in range(self.ndim)], dtype=np.float64)
        return lo, hi

    def rvs(self, size=1):
        if isinstance(size, int):
            size = (size,)
        out = np.zeros(tuple(size) + (self.ndim,))
        for k in range(self.ndim):
            out[..., k] = self.priors_in[k].rvs(size=tuple(size))
        return out
