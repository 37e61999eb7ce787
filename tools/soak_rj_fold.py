"""Long run of hens_rj_step with the adaptation folded into the next k_rj launch against k_adapt as a launch of its own
(HENS_NO_FOLD=1), same seed: every array and counter must agree bit for bit.  usage: python tools/soak_rj_fold.py [iterations]"""
import os, subprocess, sys, tempfile
import numpy as np

WORKER = r"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from eryn_amd.moves.tempering import make_ladder
from eryn_amd.rj import RJEngine, TemplateBranch
n_iter = int(sys.argv[3])
T, W, N, NL = 8, 1024, 500, 10
t = np.linspace(-1, 1, N); rs = np.random.RandomState(3)
y = 3.0 * np.exp(-((t + 0.2) ** 2) / 0.02) + 1.2 * np.sin(2 * np.pi * 7.3 * t + 1.0) + 2.0 * rs.randn(N)
brs = [TemplateBranch("gauss", "pulse", [(2.5, 3.5), (-1.0, 1.0), (0.01, 0.21)], NL, 0),
       TemplateBranch("sine", "sine", [(0.5, 1.5), (1.0, 20.0), (0.0, 2 * np.pi)], NL, 0)]
eng = RJEngine(T, W, brs, t, y, 2.0, seed=9)
x = {"gauss": np.zeros((T, W, NL, 3)), "sine": np.zeros((T, W, NL, 3))}
inds = {k: np.zeros((T, W, NL), dtype=bool) for k in x}
x["gauss"][:, :, 0] = [3.0, -0.2, 0.1] + 1e-2 * rs.randn(T, W, 3) * [1, 1, 0.1]; inds["gauss"][:, :, 0] = True
x["sine"][:, :, 0] = [1.2, 7.3, 1.0] + 1e-2 * rs.randn(T, W, 3); inds["sine"][:, :, 0] = True
eng.upload(x, inds, betas=make_ladder(6, ntemps=T)); eng.eval_state()
eng.set_mh_scale(np.full((2, 3), 1e-2) * [[1, 1, 0.1], [1, 1, 1]])
done = 0
for n in (1, 777, 64, 5000):
    while done < n_iter and n:
        k = min(n, n_iter - done); eng.step(k); done += k
        if n != 5000: break
eng.synchronize()
x1, inds1, L1, P1, betas1 = eng.download()
c = eng.counters()
np.savez(sys.argv[2], xg=x1["gauss"], xs=x1["sine"], ig=inds1["gauss"], js=inds1["sine"], L=L1, P=P1, betas=betas1,
         acc_mh=c["accepted_mh"], acc_bd=c["accepted_bd"], swaps_total=c["swaps_total"], swaps_last=c["swaps_last"])
print("done", done, "mean leaves", (inds1["gauss"].sum() + inds1["sine"].sum()) / (T * W))
"""
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
outs = []
with tempfile.TemporaryDirectory() as td:
    for env in ({}, {"HENS_NO_FOLD": "1"}):
        out = os.path.join(td, f"{len(outs)}.npz")
        e = dict(os.environ, **env)
        if not env:
            e.pop("HENS_NO_FOLD", None)
        r = subprocess.run([sys.executable, "-c", WORKER, root, out, str(n_iter)], env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        print(("folded:   " if not env else "k_adapt:  ") + r.stdout.strip().splitlines()[-1])
        outs.append(dict(np.load(out)))
bad = [k for k in outs[0] if not np.array_equal(outs[0][k], outs[1][k], equal_nan=True)]
print(f"{n_iter} iterations, 8 x 1024 walkers: " + ("bit-identical in every array and counter" if not bad else f"DIFFER in {bad}"))
sys.exit(1 if bad else 0)
