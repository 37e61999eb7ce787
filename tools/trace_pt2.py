"""Phase stamps of the stand-alone cascade k_pt_cascade (HENS_NO_FUSED=1 steps with copying launches + this kernel):
   HENS_NO_FUSED=1 python tools/trace_pt2.py T W D"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from tools.time_pipeline import make
from eryn_amd import _lib
T, W, D = map(int, sys.argv[1:4])
e = make(T, W, D)
e.step(20); e.synchronize()
acc = []
for rep in range(5):
    _lib.check(e.lib.hens_debug_trace(e.ctx, 2, None, 0, None), e.ctx)
    e.step(1); e.synchronize()
    n = 8 * 8192
    out = np.zeros(n, dtype=np.uint64); nout = C.c_int64(0)
    _lib.check(e.lib.hens_debug_trace(e.ctx, 0, _lib.ptr(out), n, C.byref(nout)), e.ctx)
    tr = out.reshape(-1, 8).astype(np.int64)[:, :7]
    tr = tr[(tr[:, 0] > 0) & (tr[:, 6] > 0)]
    acc.append(np.diff(tr, axis=1).mean(0))
    span = tr[:, 6].max() - tr[:, 0].min()
    starts = tr[:, 0] - tr[:, 0].min()
d = np.mean(acc, axis=0)
print(f"k_pt_cascade {T} x {W}: workgroups {len(tr)}  phases [phase 1 (slots, records, log u), -, barrier, walk, barrier, stores + counts]", " ".join(f"{v:6.0f}" for v in d), f" lifetime {d.sum():6.0f}  span {span}  start spread p50 {int(np.median(starts))} max {int(starts.max())}")
