"""Where the fixed cost of a short hens_step call sits: the launches' own begin / end timestamps (per-kernel events) of calls of K
iterations made from an idle, synchronised device - gaps and durations of the call's first launches against its steady state -
next to the host's wall time for the same call."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from tools.quick_bench import problem, ladder
import os; os.environ.pop("HENS_STEP_EVENTS", None)      # (quick_bench sets it on import: an event pair per call, and the HIP stream)
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood

T, W, D = 16, 4096, 32
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mu, invcov, cov = problem(D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T))
eng.eval_state(); eng.step(200); eng.synchronize()
eng.set_profiling(True)
tl, walls = [], []
for _ in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(K)            # (with per-kernel events hens_step synchronises at its end)
    walls.append((time.perf_counter() - t0) * 1e6)
    tl.append(eng.launch_times())
eng.set_profiling(False)
tl = np.median(np.array(tl[5:]), axis=0)
dur = tl[:, 1] - tl[:, 0]
gap = tl[1:, 0] - tl[:-1, 1]
print(f"K = {K}: span first begin -> last end {tl[-1, 1]:.1f} us, host wall (step incl. its sync) {np.median(walls[5:]):.1f} us")
print("launch durations (us):", np.round(dur, 2))
print("gaps (us):            ", np.round(gap, 2))
print(f"first iteration {tl[2, 0] - tl[0, 0]:.2f} us, second {tl[4, 0] - tl[2, 0]:.2f}, median of the rest {np.median(tl[6::2, 0] - tl[4:-2:2, 0]):.2f}")
