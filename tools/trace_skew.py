"""Debug: start / end skew of the workgroups of one launch on the wall clock all XCDs share (10 ns ticks).

Needs the library built with -DHENS_TRACE_REALTIME (the default stamps are the per-XCD shader clock):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DHENS_TRACE_REALTIME -Iinclude \
          eryn_amd/csrc/hens.hip -o eryn_amd/lib/libhens_rt.so
    HENS_LIB=$PWD/eryn_amd/lib/libhens_rt.so python tools/trace_skew.py [T W D]
"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from tools.quick_bench import problem, ladder
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood
from eryn_amd import _lib

T, W, D = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 4096, 32)))
mu, invcov, cov = problem(D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T))
eng.eval_state()
eng.step(200)
eng.synchronize()
for mode, name in ((1, "first half-step launch"), (3, "second half-step + cascade launch")):
    res = []
    for rep in range(5):
        _lib.check(eng.lib.hens_debug_trace(eng.ctx, mode, None, 0, None), eng.ctx)
        eng.step(1)
        eng.synchronize()
        n = T * ((W + 63) // 64) * 8
        out = np.zeros(n, dtype=np.uint64)
        nout = C.c_int64(0)
        _lib.check(eng.lib.hens_debug_trace(eng.ctx, 0, _lib.ptr(out), n, C.byref(nout)), eng.ctx)
        tr = out.reshape(-1, 8).astype(np.int64)
        tr = tr[(tr[:, 0] > 0) & (tr[:, 7] > 0)]
        t0 = tr[:, 0].min()
        res.append((len(tr), (tr[:, 7].max() - t0) / 100.0, np.percentile(tr[:, 0] - t0, [50, 90, 100]) / 100.0,
                    np.percentile(tr[:, 7] - t0, [0, 10, 50, 90]) / 100.0, np.median(tr[:, 7] - tr[:, 0]) / 100.0,
                    [float(np.median(tr[:, i + 1] - tr[:, i])) / 100.0 for i in range(7)]))
    print(name)
    for r in res:
        print(f"  workgroups {r[0]}  span first start -> last end {r[1]:.2f} us   starts (50/90/max) {np.round(r[2], 2)} us   "
              f"ends (min/10/50/90) {np.round(r[3], 2)} us   median lifetime {r[4]:.2f} us")
    print("  median phase durations (us):", np.round(res[-1][5], 2))
    d = np.diff(tr, axis=1) / 100.0
    print("  phase durations p10/p50/p90/max (us):")
    for i in range(7):
        print("    phase", i + 1, np.round(np.percentile(d[:, i], [10, 50, 90, 100]), 2))
    life = (tr[:, 7] - tr[:, 0]) / 100.0
    wg = np.arange(len(tr))
    print("  lifetime by XCD (workgroup % 8): median", np.round([np.median(life[wg % 8 == x]) for x in range(8)], 2),
          " max", np.round([life[wg % 8 == x].max() for x in range(8)], 2))
    print("  end time by XCD: median", np.round([np.median((tr[:, 7] - t0)[wg % 8 == x]) / 100.0 for x in range(8)], 2))
    slow = np.argsort(life)[-16:]
    print("  the 16 slowest workgroups:", sorted(slow.tolist()), " their phase durations (median):", np.round(np.median(d[slow], axis=0), 2))
