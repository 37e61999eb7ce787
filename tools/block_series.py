"""Block times of bench.py's timed region from a cold start: how long until blocks of 20 iterations reach their steady value?"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood
from eryn_amd.moves.tempering import make_ladder
T, W, D = 16, 4096, 32
mu, invcov = bench.gaussian_problem(D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=make_ladder(D, ntemps=T)); eng.eval_state(); eng.step(5); eng.synchronize(); eng.reset_counters()
ts = []
t_start = time.perf_counter()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.step(20); eng.synchronize(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
print("blocks (us):", np.round(ts, 1).tolist())
print(f"elapsed {1e3 * (time.perf_counter() - t_start):.1f} ms; median of first 5: {np.median(ts[:5]):.1f}, of blocks 20-40: {np.median(ts[20:40]):.1f}, last 10: {np.median(ts[-10:]):.1f}")
