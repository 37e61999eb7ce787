"""State / Branch containers with the reference's field names (eryn/state.py:330-562).

Only what the stretch + PT path touches is mirrored: one or more named branches of
``coords[ntemps, nwalkers, nleaves_max, ndim]`` with boolean ``inds``, the
``log_like`` / ``log_prior`` ``[ntemps, nwalkers]`` arrays, ``betas`` and ``random_state``.
``BranchSupplemental`` payload carriers are out of scope (None on this path).
"""
from copy import deepcopy

import numpy as np

__all__ = ["Branch", "State", "DeviceBranch", "DeviceState"]


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


# ---- the device-resident State mirror (round 6) -------------------------------------------------------------
class DeviceBranch:
    """A Branch whose ``coords`` live on the device until somebody reads them (state.py:330-395's fields)."""

    def __init__(self, owner, name, shape, inds):
        self._owner, self._name = owner, name
        self.ntemps, self.ntrees, self.nleaves_max, self.ndim = self.shape = tuple(shape)
        self.inds = inds
        self.branch_supplemental = None

    @property
    def coords(self):
        return self._owner._fetch()["x"][self._name]

    @coords.setter
    def coords(self, value):
        self._owner._fetch()["x"][self._name] = value

    @property
    def nleaves(self):
        return np.sum(self.inds, axis=-1)

    def __deepcopy__(self, memo):
        return Branch(np.array(self.coords, copy=True), inds=np.array(self.inds, copy=True))


class DeviceState(State):
    """What a device move's ``propose()`` returns when it does not copy the walkers back: the reference's State fields
    (state.py:397-562) - ``branches[name].coords``, ``log_like``, ``log_prior`` - are read from the device the FIRST time any of
    them is read (one ``hens_download_state``), ``betas`` and ``random_state`` are plain values.  A sampler loop that reads the
    state at stored steps only (``Backend.save_step``, backends/backend.py:1014-1091) therefore copies 8 (D + 2) T W bytes per
    STORED step instead of per proposal, and the move that receives the object back skips the upload as long as nobody has
    read (and so could have modified) its arrays.

    Validity: the object mirrors the device context as it stood when the move returned it.  Once the context has stepped on,
    an unread DeviceState is stale - reading it raises instead of returning the wrong iteration's walkers.  Read (or
    ``materialize()``) a state you mean to keep before the next proposal; everything already read stays valid forever."""

    def __init__(self, engine, epoch, name, shape, inds, betas=None, random_state=None):
        self._engine, self._epoch, self._name = engine, epoch, name
        self._data = None
        self._branches = {name: DeviceBranch(self, name, shape, inds)}
        self.betas = betas
        self.random_state = random_state
        self.blobs = None
        self.supplemental = None

    # -- laziness ----------------------------------------------------------------------------------
    @property
    def materialized(self):
        return self._data is not None

    def is_current(self, engine):
        """True while the device context still holds exactly this state."""
        return engine is self._engine and getattr(engine, "state_epoch", None) == self._epoch

    def _fetch(self):
        if self._data is None:
            if not self.is_current(self._engine):
                raise RuntimeError("this DeviceState was never read and its device context has stepped on since: read (or materialize()) "
                                   "a state you want to keep before the next proposal")
            x, L, P, _ = self._engine.download()
            eng = self._engine
            eng.lazy_downloads = getattr(eng, "lazy_downloads", 0) + 1
            self._data = {"x": {self._name: x[:, :, None, :]}, "L": L, "P": P}
        return self._data

    def materialize(self):
        self._fetch()
        return self

    @property
    def branches(self):
        return self._branches

    @property
    def log_like(self):
        return self._fetch()["L"]

    @log_like.setter
    def log_like(self, v):
        self._fetch()["L"] = v

    @property
    def log_prior(self):
        return self._fetch()["P"]

    @log_prior.setter
    def log_prior(self, v):
        self._fetch()["P"] = v

    def __deepcopy__(self, memo):
        d = self._fetch()
        br = self._branches[self._name]
        return State({self._name: np.array(d["x"][self._name], copy=True)}, inds={self._name: np.array(br.inds, copy=True)},
                     log_like=np.array(d["L"], copy=True), log_prior=np.array(d["P"], copy=True),
                     betas=None if self.betas is None else np.array(self.betas, copy=True), random_state=deepcopy(self.random_state, memo))
