"""Dev build with -DHENS_TRACE_WAVES: arrival of each wave of the FIRST launch (k_stretch_fast) at the first barrier."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from tools.quick_bench import problem, ladder
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood
from eryn_amd import _lib
T, W, D = 16, 4096, 32
mu, invcov, cov = problem(D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T)); eng.eval_state(); eng.step(200); eng.synchronize()
_lib.check(eng.lib.hens_debug_trace(eng.ctx, 1, None, 0, None), eng.ctx)
eng.step(1); eng.synchronize()
n = T * ((W + 63) // 64) * 8
out = np.zeros(n, dtype=np.uint64); nout = C.c_int64(0)
_lib.check(eng.lib.hens_debug_trace(eng.ctx, 0, _lib.ptr(out), n, C.byref(nout)), eng.ctx)
tr = out.reshape(-1, 8).astype(np.int64)
tr = tr[(tr[:, 0] > 0) & (tr[:, 1] > 0) & (tr[:, 2] > 0)]
names = {1: "w1 adaptation", 2: "w2 complement row", 3: "w3", 7: "w0 own record + zz"}
for i, nm in names.items():
    rel = tr[:, i] - tr[:, 0]
    print(f"{nm:20s} arrival at barrier 1, cycles after start: mean {rel.mean():7.0f}  p10 {np.percentile(rel, 10):6.0f}  p50 {np.percentile(rel, 50):6.0f}  p90 {np.percentile(rel, 90):6.0f}")
