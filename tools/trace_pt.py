"""Debug: per-phase timing of the PT cascade kernel from in-kernel s_memtime stamps."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from tools.quick_bench import problem, ladder
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood
from eryn_amd import _lib

T, W, D = 16, 4096, 32
mu, invcov, cov = problem(D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T))
eng.eval_state()
eng.step(200)
eng.synchronize()
_lib.check(eng.lib.hens_debug_trace(eng.ctx, 2, None, 0, None), eng.ctx)
eng.step(1)
eng.synchronize()
n = max(T * ((W + 63) // 64), (W + 15) // 16) * 8
out = np.zeros(n, dtype=np.uint64)
nout = C.c_int64(0)
_lib.check(eng.lib.hens_debug_trace(eng.ctx, 0, _lib.ptr(out), n, C.byref(nout)), eng.ctx)
tr = out.reshape(-1, 8).astype(np.int64)[:, :7]
tr = tr[(tr[:, 0] > 0) & (tr[:, 6] > 0)]
names = ["keys+bar", "phase1 (prp, gathers, log)", "bar", "walk", "bar", "phase3 stores"]
d = np.diff(tr, axis=1)
print("workgroups:", len(tr), "lifetime mean", (tr[:, 6] - tr[:, 0]).mean())
print(dict(zip(names, np.round(d.mean(0), 1))))
