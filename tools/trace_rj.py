"""Phase stamps of k_rj (config 4): python tools/trace_rj.py [4 = in-model move | 5 = birth / death]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
import bench
from eryn_amd import _lib
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 5
class A: ntemps = None; nwalkers = None; warmup = 50; steps = 1
# build the config-4 engine exactly like bench.run_cfg4, but keep it
import types
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
eng.step(50); eng.synchronize()
e = eng.eng if hasattr(eng, "eng") else eng
_lib.check(e.lib.hens_debug_trace(e.ctx, mode, None, 0, None), e.ctx)
eng.step(1); eng.synchronize()
n = T * ((W + 63) // 64) * 8
out = np.zeros(n, dtype=np.uint64); nout = C.c_int64(0)
_lib.check(e.lib.hens_debug_trace(e.ctx, 0, _lib.ptr(out), n, C.byref(nout)), e.ctx)
tr = out.reshape(-1, 8)[:, :6].astype(np.int64); tr = tr[(tr[:, 0] > 0) & (tr[:, 5] > 0)]
d = np.diff(tr, axis=1)
print("waves traced", len(tr), "mode", mode, " phases: load, proposal, log-prior, likelihood, accept+update")
print("mean", np.round(d.mean(0), 0), " median", np.median(d, axis=0), " lifetime mean", (tr[:, 5] - tr[:, 0]).mean())
# (with a library built -DHENS_RJ_TRACE_STRIDE the traced waves are every 64th walker: quarters of the launch in dispatch order)
q = len(tr) // 4
for k in range(4):
    dd = d[k * q:(k + 1) * q]
    print(f"  quarter {k}: phases mean", np.round(dd.mean(0), 0), " lifetime mean", round(float((tr[k * q:(k + 1) * q, 5] - tr[k * q:(k + 1) * q, 0]).mean())), "p90", round(float(np.percentile(tr[k * q:(k + 1) * q, 5] - tr[k * q:(k + 1) * q, 0], 90))))
if "raw" in sys.argv:
    for k in (0, 1, 2, 63, 64, 65, 127, 128, 129, 130, 200, 255):
        print(k, d[k], tr[k, 0] - tr[:, 0].min())
if "rungs" in sys.argv:          # per rung: 32 traced waves each (stride build); start offsets are only comparable within an XCD
    per = len(tr) // T
    for r in range(T):
        sl = slice(r * per, (r + 1) * per)
        st = tr[sl, 0] - tr[:, 0].min()
        print(f"  rung {r}: phases", np.round(d[sl].mean(0)), "lifetime", round(float((tr[sl, 5] - tr[sl, 0]).mean())), "start offset min/med/max", int(st.min()), int(np.median(st)), int(st.max()), "end max", int((tr[sl, 5] - tr[:, 0].min()).max()))
