"""Config 4: active leaves per rung after the bench's warm-up, and the iteration time.   python tools/probe/rj_rung_leaves.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from eryn_amd.moves.tempering import make_ladder
from eryn_amd.rj import RJEngine, TemplateBranch
T, W, N, NL = 8, 2048, 500, 10
t = np.linspace(-1, 1, N); rs = np.random.RandomState(42)
gauss_inj = np.array([[3.3, -0.2, 0.1], [2.6, -0.1, 0.1], [3.4, 0.0, 0.1], [2.9, 0.3, 0.1]]); sine_inj = np.array([[1.3, 10.1, 1.0], [0.8, 4.6, 1.2]])
y = sum(a * np.exp(-((t - b) ** 2) / (2 * c ** 2)) for a, b, c in gauss_inj) + sum(a * np.sin(2 * np.pi * b * t + c) for a, b, c in sine_inj) + 2.0 * rs.randn(N)
brs = [TemplateBranch("gauss", "pulse", [(2.5, 3.5), (-1.0, 1.0), (0.01, 0.21)], NL, 0), TemplateBranch("sine", "sine", [(0.5, 1.5), (1.0, 20.0), (0.0, 2 * np.pi)], NL, 0)]
eng = RJEngine(T, W, brs, t, y, 2.0, seed=2024)
x = {"gauss": np.zeros((T, W, NL, 3)), "sine": np.zeros((T, W, NL, 3))}; inds = {k: np.zeros((T, W, NL), dtype=bool) for k in x}
for n in range(4): x["gauss"][:, :, n] = gauss_inj[n] + 1e-2 * rs.randn(T, W, 3) * [1, 1, 0.1]; inds["gauss"][:, :, n] = True
for n in range(2): x["sine"][:, :, n] = sine_inj[n] + 1e-2 * rs.randn(T, W, 3); inds["sine"][:, :, n] = True
eng.upload(x, inds, betas=make_ladder(18, ntemps=T)); eng.eval_state(); eng.set_mh_scale(np.full((2, 3), 1e-2) * [[1, 1, 0.1], [1, 1, 1]])
eng.step(100); eng.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); eng.step(1000); eng.synchronize(); dt = time.perf_counter() - t0
    st = eng.download()
    inds1 = st[1]
    per = {k: v.reshape(T, W, -1).sum(-1).mean(1) for k, v in inds1.items()}
    print(f"{dt / 1000 * 1e6:7.2f} us/iter   leaves per rung: " + "  ".join(f"{k} " + " ".join(f"{v:.2f}" for v in p) for k, p in per.items()), flush=True)
