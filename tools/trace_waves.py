"""Dev build with -DHENS_TRACE_WAVES: when does each wave of k_split1_pt reach the first barrier?  (tools/devbuild.sh waves -DHENS_TRACE_WAVES)"""
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
_lib.check(eng.lib.hens_debug_trace(eng.ctx, 3, None, 0, None), eng.ctx)
eng.step(1); eng.synchronize()
n = T * ((W + 63) // 64) * 8
out = np.zeros(n, dtype=np.uint64); nout = C.c_int64(0)
_lib.check(eng.lib.hens_debug_trace(eng.ctx, 0, _lib.ptr(out), n, C.byref(nout)), eng.ctx)
tr = out.reshape(-1, 8).astype(np.int64)
tr = tr[(tr[:, 0] > 0) & (tr[:, 7] > 0)]
rel = tr[:, 1:] - tr[:, :1]
names = ["w1 slots", "w2 uniforms", "w3 uniforms", "w4 betas (+scol)", "w5 complement row", "w6 zz/logs", "w7 (scol)"]
for i, nm in enumerate(names):
    print(f"{nm:20s} arrival at barrier 1, cycles after start: mean {rel[:, i].mean():7.0f}  p10 {np.percentile(rel[:, i], 10):6.0f}  p50 {np.percentile(rel[:, i], 50):6.0f}  p90 {np.percentile(rel[:, i], 90):6.0f}")
late = rel.argmax(axis=1)
print("last wave to arrive (share of workgroups):", {names[k]: round(float((late == k).mean()), 2) for k in range(7)})
