"""Phase stamps of both launches (stretch move only), Rosenbrock or - 4th argument `dense` - the Gaussian likelihood:
  python tools/trace_fused2.py T W D [dense]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import RosenbrockLikelihood
from eryn_amd.moves.tempering import make_ladder
from eryn_amd import _lib
T, W, D = map(int, sys.argv[1:4])
if len(sys.argv) > 4 and sys.argv[4] == "dense":
    from tools.quick_bench import problem, ladder
    from eryn_amd.likelihood import GaussianLikelihood
    mu, invcov, cov = problem(D)
    e = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
    e.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T))
else:
    e = HipEnsemble(T, W, D, RosenbrockLikelihood(D), -5.0, 5.0, seed=2024)
    e.upload(np.clip(1.0 + 0.05 * np.random.RandomState(1).randn(T, W, D), -4.9, 4.9), betas=make_ladder(D, ntemps=T))
e.eval_state(); e.step(50); e.synchronize()
for which, label in ((1, "k_stretch_fast"), (3, "k_split1_pt")):
    acc = []
    for rep in range(3):
        _lib.check(e.lib.hens_debug_trace(e.ctx, which, None, 0, None), e.ctx)
        e.step(1); e.synchronize()
        n = 8 * 8192
        out = np.zeros(n, dtype=np.uint64); nout = C.c_int64(0)
        _lib.check(e.lib.hens_debug_trace(e.ctx, 0, _lib.ptr(out), n, C.byref(nout)), e.ctx)
        tr = out.reshape(-1, 8).astype(np.int64)
        tr = tr[(tr[:, 0] > 0) & (tr[:, 7] > 0)]
        acc.append(np.diff(tr, axis=1).mean(0))
    d = np.mean(acc, axis=0)
    print(f"{label:16s} {T} x {W} x {D}: workgroups {len(tr)}  phases", " ".join(f"{v:6.0f}" for v in d), f" lifetime {d.sum():6.0f}", flush=True)
e.close()
