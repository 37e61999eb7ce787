"""Host-side cost of the calls inside bench.py's timed region when the GPU has nothing to do (us per call)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood
T, W, D = 16, 4096, 32
mu, invcov = bench.gaussian_problem(D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=np.geomspace(1, 0.01, T)); eng.eval_state(); eng.step(100); eng.synchronize()
def t(f, n=2000):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print(f"eng.step(0)                 {t(lambda: eng.step(0)):6.2f} us")
print(f"eng.synchronize() idle      {t(eng.synchronize):6.2f} us")
print(f"torch.cuda.synchronize()    {t(torch.cuda.synchronize):6.2f} us")
print(f"torch._C._cuda_synchronize  {t(torch._C._cuda_synchronize):6.2f} us")
print(f"time.perf_counter()         {t(time.perf_counter):6.2f} us")
lib, ctx = eng.lib, eng.ctx
print(f"raw lib.hens_step(ctx, 0)   {t(lambda: lib.hens_step(ctx, 0)):6.2f} us")
print(f"raw lib.hens_synchronize    {t(lambda: lib.hens_synchronize(ctx)):6.2f} us")
for K in (1, 20):
    ts = []
    for _ in range(200):
        torch.cuda.synchronize(); t0 = time.perf_counter(); eng.step(K); eng.synchronize(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    a = np.median(np.array(ts), axis=0) * 1e6
    print(f"K = {K}: step + eng.synchronize {a[0]:.1f} us, then torch.cuda.synchronize {a[1]:.1f} us")
