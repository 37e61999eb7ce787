"""Phase stamps of both stepping launches, the shape alone and as a (lone) pipeline rank:  python tools/trace_pipe_phases.py T W D
   (s_memtime ticks; single-GPU contexts: HENS_NO_AQL=1 is set by this script - traces go through the HIP stream anyway)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from tools.time_pipeline import make
from eryn_amd.ladder import LadderPipeline
from eryn_amd import _lib

T, W, D = map(int, sys.argv[1:4])
for which, label in ((1, "first launch (k_stretch_fast)"), (3, "second launch (k_split1_pt)")):
    for kind in ("single", "pipe"):
        e = make(T, W, D, (0, T) if kind == "pipe" else None)
        if kind == "pipe":
            LadderPipeline.connect_local([e])
        e.step(100); e.synchronize()
        acc = []
        for rep in range(5):
            _lib.check(e.lib.hens_debug_trace(e.ctx, which, None, 0, None), e.ctx)
            e.step(3); e.synchronize()
            n = 8 * 4096
            out = np.zeros(n, dtype=np.uint64); nout = C.c_int64(0)
            _lib.check(e.lib.hens_debug_trace(e.ctx, 0, _lib.ptr(out), n, C.byref(nout)), e.ctx)
            tr = out.reshape(-1, 8).astype(np.int64)
            wg0 = tr[0].copy()
            tr = tr[(tr[:, 0] > 0) & (tr[:, 7] > 0)]
            acc.append(np.diff(tr, axis=1).mean(0))
            span = tr[:, 7].max() - tr[:, 0].min()
        d = np.mean(acc, axis=0)
        life = tr[:, 7] - tr[:, 0]
        print(f"{label:32s} {kind:6s} workgroups {len(tr):5d}  phases", " ".join(f"{v:7.0f}" for v in d), f" lifetime {d.sum():7.0f}  max {life.max()}  p99 {int(np.percentile(life, 99))}  workgroup (0,0): phases", " ".join(str(int(v)) for v in np.diff(wg0)), f"lifetime {wg0[7] - wg0[0]}", flush=True)
        e.close()
