"""State / Branch containers with the reference's field names (eryn/state.py:330-562).

Only what the stretch + PT path touches is mirrored: one or more named branches of
``coords[ntemps, nwalkers, nleaves_max, ndim]`` with boolean ``inds``, the
``log_like`` / ``log_prior`` ``[ntemps, nwalkers]`` arrays, ``betas`` and ``random_state``.
``BranchSupplemental`` payload carriers are out of scope (None on this path).
"""
from copy import deepcopy

import numpy as np


class Branch:
    """One model branch (state.py:330-395)."""

    def __init__(self, coords, inds=None, branch_supplemental=None):
        if coords.ndim != 4:
            raise ValueError("Branch coords must be 4-D (ntemps, nwalkers, nleaves_max, ndim).")
        self.coords = coords
        self.ntemps, self.ntrees, self.nleaves_max, self.ndim = self.shape = coords.shape
        if inds is None:
            inds = np.full(coords.shape[:3], True)
        elif not isinstance(inds, np.ndarray):
            raise ValueError("inds must be np.ndarray in Branch.")
        elif inds.shape != coords.shape[:3]:
            raise ValueError("inds has wrong shape.")
        self.inds = inds
        if branch_supplemental is not None:
            raise NotImplementedError("branch supplementals are outside the device hot path")
        self.branch_supplemental = None

    @property
    def nleaves(self):
        return np.sum(self.inds, axis=-1)


class State:
    """Sampler state (state.py:397-562): same constructor conventions as the reference."""

    def __init__(self, coords, inds=None, branch_supplemental=None, supplemental=None, log_like=None,
                 log_prior=None, betas=None, blobs=None, random_state=None, copy=False):
        dc = deepcopy if copy else (lambda v: v)
        if hasattr(coords, "branches"):                # another State
            for f in ("branches", "log_like", "log_prior", "blobs", "betas", "supplemental", "random_state"):
                setattr(self, f, dc(getattr(coords, f)))
            return
        if isinstance(coords, np.ndarray):
            coords = {"model_0": coords}
        elif not isinstance(coords, dict):
            raise ValueError("Input coords need to be np.ndarray, dict, or State object.")
        coords = dict(coords)
        for name, arr in coords.items():
            if arr.ndim == 2:                          # (nwalkers, ndim)
                arr = arr[None, :, None, :]
            elif arr.ndim == 3:                        # (ntemps, nwalkers, ndim)
                arr = arr[:, :, None, :]
            elif arr.ndim != 4:
                raise ValueError(f"Dimension of coordinates must be between 2 and 4. coords dimension is {arr.ndim}.")
            coords[name] = arr
        if inds is None:
            inds = {k: None for k in coords}
        elif not isinstance(inds, dict):
            raise ValueError("inds must be None or dict.")
        if branch_supplemental is not None and any(v is not None for v in branch_supplemental.values()):
            raise NotImplementedError("branch supplementals are outside the device hot path")
        self.branches = {k: Branch(dc(v), inds=inds.get(k)) for k, v in coords.items()}
        self.log_like = dc(np.atleast_2d(log_like)) if log_like is not None else None
        self.log_prior = dc(np.atleast_2d(log_prior)) if log_prior is not None else None
        self.blobs = dc(np.atleast_3d(blobs)) if blobs is not None else None
        self.betas = dc(np.atleast_1d(betas)) if betas is not None else None
        self.supplemental = dc(supplemental)
        self.random_state = dc(random_state)

    @property
    def branches_inds(self):
        return {k: b.inds for k, b in self.branches.items()}

    @property
    def branches_coords(self):
        return {k: b.coords for k, b in self.branches.items()}

    @property
    def branches_supplemental(self):
        return {k: b.branch_supplemental for k, b in self.branches.items()}

    @property
    def branch_names(self):
        return list(self.branches.keys())

    def get_log_posterior(self, temper=False):
        """log posterior [ntemps, nwalkers]; betas broadcast along the temperature axis."""
        betas = self.betas if (temper and self.betas is not None) else np.ones(self.log_like.shape[0])
        return betas[:, None] * self.log_like + self.log_prior
