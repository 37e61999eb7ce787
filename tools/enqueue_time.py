import sys, time, numpy as np
sys.path.insert(0, ".")
from tools.quick_bench import problem, ladder
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood
T, W, D = 16, 4096, 32
mu, invcov, cov = problem(D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T)); eng.eval_state()
eng.step(200); eng.synchronize()
for n in (2000, 2000, 4000):
    t0 = time.perf_counter(); eng.step(n); t1 = time.perf_counter(); eng.synchronize(); t2 = time.perf_counter()
    print(f"n={n}: enqueue {1e6*(t1-t0)/n:.2f} us/iter, total {1e6*(t2-t0)/n:.2f} us/iter")
