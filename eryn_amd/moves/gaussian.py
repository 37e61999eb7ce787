"""HIP-backed ``GaussianMove`` (a Metropolis-Hastings move with a Gaussian proposal) with the reference's
plugin contract: ``MHMove.propose`` (mh.py:56-193) + ``GaussianMove.get_proposal`` (gaussian.py:68-195) for
the single-branch ``nleaves_max == 1`` case.  The proposal draws come from ``model.random`` in the
reference's order (gaussian.py:161-176, 265-268; mh.py:157), the step ``q - x`` is formed on the host from
those draws and everything else - q = x + step, box prior, likelihood, tempered accept test, update, PT
sweep - runs in libhipensemble (``hens_mh_step``).  With the same seeds the chain is the reference's.
"""
import numpy as np

from .device import DeviceMove

__all__ = ["GaussianMove", "MHMove"]


class MHMove(DeviceMove):
    """Base of the full-ensemble Metropolis-Hastings moves: subclasses return the step ``q - x`` of every
    walker from ``get_step(random, n, ndim)`` (the device adds it, so factors = 0: symmetric proposals)."""

    def get_step(self, random, n, ndim):
        raise NotImplementedError("The proposal must be implemented by subclasses")

    def propose(self, model, state):
        name, br, T, W, D = self._single_branch(state)
        eng = self._ensure_engine(T, W, D)
        if hasattr(eng.likelihood, "evaluate"):
            raise NotImplementedError("MH moves on the device need a device likelihood")
        if self.rng == "philox":                                       # device-side normals (k_stretch_fast<MODE_MH> / k_mh_draw)
            return self._propose_philox(model, state, mh_proposal=self.device_proposal())
        self._apply_periodic(eng, name, D)
        self._upload_if_needed(eng, state, br)
        self._bump(eng)
        step = self.get_step(model.random, T * W, D)                   # gaussian.py:116 (all leaves active)
        u_acc = model.random.rand(T, W)                                # mh.py:157
        accepted = eng.mh_step(step, u_acc)
        if self._accepted is not None:
            self.accepted += accepted                                  # mh.py:186
        self.num_proposals += 1
        return self._finish(eng, state, name, br, T, W), accepted


class GaussianMove(MHMove):
    """``GaussianMove(cov_all, mode="vector", factor=None)`` (gaussian.py:9-66).

    ``cov_all``: ``{branch_name: cov}`` with ``cov`` a scalar (isotropic), a 1-D array (axis-aligned; the
    reference itself cannot construct this case at current numpy - gaussian.py:144 - so it has no parity
    fixture) or a square matrix (general, ``vector`` mode only, drawn with ``random.multivariate_normal``).
    """

    def __init__(self, cov_all, mode="vector", factor=None, **kwargs):
        if len(cov_all) != 1:
            raise NotImplementedError("the device path handles a single branch")
        (self.branch_name, cov), = cov_all.items()
        try:
            float(cov)
        except TypeError:
            cov = np.atleast_1d(np.asarray(cov, dtype=np.float64))
            if cov.ndim == 1:
                self.kind, self.scale = "diag", np.sqrt(cov)           # gaussian.py:48-50
            elif cov.ndim == 2 and cov.shape[0] == cov.shape[1]:
                self.kind, self.scale = "full", cov                    # gaussian.py:51-54
                if mode != "vector":                                   # gaussian.py:260
                    raise ValueError(f"'{mode}' is not a recognized mode. Please select from: ['vector']")
            else:
                raise ValueError("Invalid proposal scale dimensions")
        else:
            self.kind, self.scale = "iso", np.sqrt(cov)                # gaussian.py:58-60
        if factor is not None and factor < 1.0:
            raise ValueError("'factor' must be >= 1.0")                # gaussian.py:149-150
        if mode not in ("vector", "random", "sequential"):
            raise ValueError(f"'{mode}' is not a recognized mode. Please select from: ['vector', 'random', 'sequential']")
        self.mode = mode
        self._log_factor = None if factor is None else np.log(factor)
        self.index = 0
        MHMove.__init__(self, **kwargs)

    def get_factor(self, rng):                                         # gaussian.py:161-164
        if self._log_factor is None:
            return 1.0
        return np.exp(rng.uniform(-self._log_factor, self._log_factor))

    def get_step(self, random, n, ndim):
        f = self.get_factor(random)
        if self.kind == "full":                                        # gaussian.py:265-268
            step = f * random.multivariate_normal(np.zeros(len(self.scale)), self.scale, size=n)
        else:                                                          # gaussian.py:166-167, 255-256
            step = f * self.scale * random.randn(n, ndim)
        if self.mode == "vector":
            return step
        if self.mode == "random":                                      # gaussian.py:172-173
            m = random.randint(ndim, size=n)
        else:                                                          # gaussian.py:174-176
            m = self.index % ndim + np.zeros(n, dtype=int)
            self.index = (self.index + 1) % ndim
        out = np.zeros_like(step)
        rows = np.arange(n)
        out[rows, m] = step[rows, m]
        return out

    def device_proposal(self):
        """(kind, scale) for ``HipEnsemble.set_mh_proposal`` (device-side draws, vector mode, no factor)."""
        if self.mode != "vector" or self._log_factor is not None:
            raise NotImplementedError("device-side Gaussian draws implement mode='vector' without factor")
        if self.kind == "full":
            return "full", np.linalg.cholesky(np.asarray(self.scale, dtype=np.float64))
        return self.kind, np.atleast_1d(self.scale)
