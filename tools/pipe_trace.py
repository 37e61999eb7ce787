"""Phase stamps of a pipeline rank's FIRST launch (k_stretch_fast<PIPE>), the adapting workgroup (0,0) beside the others.
  python tools/pipe_trace.py T W D [single]      env PIPE_DELAY=0/1, PIPE_ADAPTIVE, HENS_PIPE_INJECT_CYCLES
Stamps are s_memtime (shader cycles, one counter per XCD: only durations inside a workgroup compare) - or, with a library built
-DHENS_TRACE_REALTIME (tools/devbuild.sh rt -DHENS_TRACE_REALTIME -DHENS_DEV_D=64), the 100 MHz wall clock all XCDs share:
then the start / end columns are the launch's own timeline."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from tools.time_pipeline import make  # noqa: E402
from eryn_amd.ladder import LadderPipeline  # noqa: E402
from eryn_amd import _lib  # noqa: E402

T, W, D = map(int, sys.argv[1:4])
single = "single" in sys.argv[4:]
NSTEP = int(os.environ.get("TRACE_NSTEP", "6"))
e = make(T, W, D, None if single else (0, T))
if not single:
    LadderPipeline.connect_local([e])
e.step(200)
e.synchronize()
ntile = (W // 2 + 63) // 64
names = ["start", "A done", "bar1", "B done", "bar2", "C done", "D done", "end"]
acc = []
for rep in range(5):
    _lib.check(e.lib.hens_debug_trace(e.ctx, 1, None, 0, None), e.ctx)
    e.step(NSTEP)                # (the stamps of the call's LAST iteration survive: NSTEP > 1 = a launch in the middle of a call's chain,
    e.synchronize()              #  where the previous sweep's counts are pushed by this launch's head and not by the call's epilogue)
    n = T * ((W + 63) // 64) * 8
    out = np.zeros(n, dtype=np.uint64)
    nout = C.c_int64(0)
    _lib.check(e.lib.hens_debug_trace(e.ctx, 0, _lib.ptr(out), n, C.byref(nout)), e.ctx)
    tr = out.reshape(-1, 8).astype(np.int64)[: T * ntile]      # (grid of the first launch: T rungs x W / 128 tiles, index y * gridDim.x + x)
    acc.append(tr)
    e.step(3)
tr = acc[-1]
ok = (tr[:, 0] > 0) & (tr[:, 7] > 0)
print(f"{'single' if single else 'pipe rank'} T={T} W={W} D={D} delay={os.environ.get('PIPE_DELAY', '0')} inject={os.environ.get('HENS_PIPE_INJECT_CYCLES', '0')}: "
      f"{ok.sum()} of {len(tr)} workgroups traced")
rel = tr - tr[:, :1]
oth = rel[1:][ok[1:]]
print("                 " + "".join(f"{n:>9s}" for n in names))
print("workgroup (0,0): " + "".join(f"{v:9d}" for v in rel[0]))
for q in (50, 90, 99, 100):
    print(f"others p{q:<3d}:     " + "".join(f"{int(np.percentile(oth[:, i], q)):9d}" for i in range(8)))
d = np.diff(oth, axis=1)
print("others, phase durations (median): " + ", ".join(f"{n} {int(np.median(d[:, i]))}" for i, n in enumerate(names[1:])))
t0 = tr[ok][:, 0].min()
print(f"launch timeline (meaningful with the wall clock only): first start 0, last start {tr[ok][:, 0].max() - t0}, "
      f"workgroup (0,0) ends {tr[0, 7] - t0}, median end {int(np.median(tr[ok][:, 7])) - t0}, last end {tr[ok][:, 7].max() - t0} "
      f"(workgroup {int(np.argmax(np.where(ok, tr[:, 7], 0)))})")
# the five repeats: lifetime of workgroup (0,0) and the p99 lifetime
for k, t in enumerate(acc):
    o = (t[:, 0] > 0) & (t[:, 7] > 0)
    life = (t[:, 7] - t[:, 0])[o]
    print(f"  repeat {k}: workgroup (0,0) lifetime {t[0, 7] - t[0, 0]}, others median {int(np.median(life[1:]))}, p99 {int(np.percentile(life[1:], 99))}, "
          f"phase D (C done -> D done) median {int(np.median((t[:, 6] - t[:, 5])[o][1:]))} p99 {int(np.percentile((t[:, 6] - t[:, 5])[o][1:], 99))}")
e.close()
