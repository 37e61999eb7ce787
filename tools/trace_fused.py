"""Debug: per-phase timing of the fused half-step + cascade kernel (k_split1_pt) from in-kernel s_memtime stamps."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from tools.quick_bench import problem, ladder
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood
from eryn_amd import _lib

T, W, D = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 4096, 32)))
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 3          # 3: fused kernel, 1: first half-step kernel
mu, invcov, cov = problem(D)
import os
PIPE = bool(os.environ.get("TRACE_PIPE"))              # the same shape as a 1-rank ladder pipeline (k_split1_pt<PIPE>)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024, rung_range=(0, T) if PIPE else None,
                  adaptation_delay=int(os.environ.get("PIPE_DELAY", "0")) if PIPE else 0)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T))
eng.eval_state()
if PIPE:
    from eryn_amd.ladder import LadderPipeline
    LadderPipeline.connect_local([eng])
eng.step(200)
eng.synchronize()
_lib.check(eng.lib.hens_debug_trace(eng.ctx, mode, None, 0, None), eng.ctx)
eng.step(int(sys.argv[5]) if len(sys.argv) > 5 else 2)      # 1: every stamp comes from the same launch
eng.synchronize()
n = T * ((W + 63) // 64) * 8
out = np.zeros(n, dtype=np.uint64)
nout = C.c_int64(0)
_lib.check(eng.lib.hens_debug_trace(eng.ctx, 0, _lib.ptr(out), n, C.byref(nout)), eng.ctx)
tr_all = out.reshape(-1, 8).astype(np.int64)
ok = (tr_all[:, 0] > 0) & (tr_all[:, 7] > 0)
tr = tr_all[ok]
wg = np.flatnonzero(ok)
t0 = tr[:, 0].min()
names = ["start", "A done", "bar1", "B done", "C done", "D done", "F done", "end"] if mode == 3 else \
        ["start", "A done", "bar1", "B done", "bar2", "C done", "D done", "end"]
print("workgroups traced:", len(tr), " launch span (first start -> last end):", tr[:, 7].max() - t0,
      " start spread:", tr[:, 0].max() - t0, " end spread:", tr[:, 7].max() - tr[:, 7].min())
print("start percentiles (10/50/90/99/max):", np.percentile(tr[:, 0] - t0, [10, 50, 90, 99, 100]).astype(int))
print("end percentiles (0/10/50/90/max):", np.percentile(tr[:, 7] - t0, [0, 10, 50, 90, 100]).astype(int))
for i, nm in enumerate(names):
    rel = tr[:, i] - t0
    print(f"{nm:8s} mean {rel.mean():9.1f}  min {rel.min():7d}  max {rel.max():7d}")
d = np.diff(tr, axis=1)
print("phase durations mean:", dict(zip(names[1:], np.round(d.mean(0), 1))))
print("wg lifetime mean", (tr[:, 7] - tr[:, 0]).mean(), "start spread", tr[:, 0].max() - t0)

print("percentiles (10/50/90/99/max) per phase:")
for i, nm in enumerate(names[1:]):
    print(f"  {nm:8s}", np.percentile(d[:, i], [10, 50, 90, 99, 100]).astype(int))
life = tr[:, 7] - tr[:, 0]
print("lifetime percentiles:", np.percentile(life, [10, 50, 90, 99, 100]).astype(int))
# the launch as seen from the first workgroup's start (drop the other iteration's stamps: keep the later launch)
late = tr[:, 0] > np.median(tr[:, 0]) if (tr[:, 0].max() - tr[:, 0].min()) > 10 * life.max() else np.ones(len(tr), bool)
tl = tr[late]
print("later launch: workgroups", len(tl), " span start->last end", tl[:, 7].max() - tl[:, 0].min(), " start spread", tl[:, 0].max() - tl[:, 0].min())
k = 6 if mode == 1 else 6
print("phase index", k, "duration by XCD (wg % 8):", [float(np.round(np.nan_to_num(d[late][(wg[late] % 8) == x, k - 1].mean()))) for x in range(8)])
print("first 8 workgroups, phase durations:")
print(d[late][:8])


