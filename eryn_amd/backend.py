"""Minimal in-memory chain store so ``run_mcmc(store=True)`` is usable.

The reference's storage engine (eryn/backends, HDF5, resume, ACT/evidence accessors) is out
of scope for this package (host-side I/O, consumes ``State`` snapshots).  This class
keeps the accessor names the stretch + PT path's callers use.
"""
import numpy as np


class Backend:
    def __init__(self):
        self.initialized = False

    def reset(self, nwalkers, ndims, ntemps=1, branch_names=None, **kwargs):
        self.nwalkers, self.ndims, self.ntemps = nwalkers, dict(ndims), ntemps
        self.branch_names = list(branch_names)
        self.iteration = 0
        self.chain = {k: np.empty((0, ntemps, nwalkers, 1, d)) for k, d in self.ndims.items()}
        self.log_like = np.empty((0, ntemps, nwalkers))
        self.log_prior = np.empty((0, ntemps, nwalkers))
        self.betas = np.empty((0, ntemps))
        self.accepted = np.zeros((ntemps, nwalkers))
        self.swaps_accepted = np.zeros(max(ntemps - 1, 0))
        self.random_state = None
        self.initialized = True

    def grow(self, ngrow, blobs=None):
        self._cap = self.iteration + ngrow
        for k in self.chain:
            a = self.chain[k]
            self.chain[k] = np.concatenate([a, np.empty((self._cap - a.shape[0],) + a.shape[1:])])
        for f in ("log_like", "log_prior", "betas"):
            a = getattr(self, f)
            setattr(self, f, np.concatenate([a, np.empty((self._cap - a.shape[0],) + a.shape[1:])]))

    def save_step(self, state, accepted, swaps_accepted=None, **kwargs):
        i = self.iteration
        for k, br in state.branches.items():
            self.chain[k][i] = br.coords
        self.log_like[i] = state.log_like
        self.log_prior[i] = state.log_prior
        if state.betas is not None:
            self.betas[i] = state.betas
        self.accepted += accepted
        if swaps_accepted is not None and len(swaps_accepted):
            self.swaps_accepted += swaps_accepted
        self.random_state = state.random_state
        self.iteration += 1

    def get_chain(self, discard=0, thin=1):
        return {k: v[discard:self.iteration:thin] for k, v in self.chain.items()}

    def get_log_like(self, discard=0, thin=1):
        return self.log_like[discard:self.iteration:thin]

    def get_log_prior(self, discard=0, thin=1):
        return self.log_prior[discard:self.iteration:thin]

    def get_betas(self, discard=0, thin=1):
        return self.betas[discard:self.iteration:thin]
