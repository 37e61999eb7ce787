"""Log-likelihood of the library's evaluation launch against numpy (extended precision) on random states: max relative error.
usage: [HENS_LIB=...] python tools/like_check.py T W D"""
import sys
import numpy as np
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood

T, W, D = (int(v) for v in sys.argv[1:4])
rs = np.random.RandomState(5)
A = rs.randn(D, D)
mu = 0.3 * rs.randn(D)
invcov = np.linalg.inv(A @ A.T / D + np.eye(D))
if len(sys.argv) > 4:                      # a NON-symmetric precision matrix: the form uses A_ik + A_ki
    invcov = invcov + 0.01 * rs.randn(D, D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=1)
x0 = rs.randn(T, W, D) * 2.0
x0[0, 0, 0] = 60.0                          # one walker outside the prior box
from eryn_amd.moves.tempering import make_ladder
eng.upload(x0, betas=make_ladder(D, ntemps=T))
eng.eval_state()
x, L, P, betas = eng.download()
d = (x0 - mu).astype(np.longdouble)
ref = -0.5 * np.einsum("twi,ij,twj->tw", d, invcov.astype(np.longdouble), d)
inb = np.isfinite(P)
rel = np.abs((L[inb] - ref[inb]) / ref[inb]).max()
print(f"{T}x{W}x{D}: max relative error of log-likelihood {float(rel):.3e}; out-of-box walker L = {L[0, 0]}, P = {P[0, 0]}")
eng.step(200); eng.synchronize()
x, L, P, betas = eng.download()
d = (x - mu).astype(np.longdouble)
ref = -0.5 * np.einsum("twi,ij,twj->tw", d, invcov.astype(np.longdouble), d)
print(f"after 200 iterations: max relative error {float(np.abs((L - ref) / ref).max()):.3e}, acceptance {eng.counters()['accepted'].mean() / 200:.3f}")
