"""Per-launch durations of a pipeline rank (1-rank pipeline: no neighbours) next to the same shape as a ladder of its own.
  python tools/pipe_prof.py T W D [iters]      env PIPE_DELAY=0/1, HENS_PIPE_NO_FUSED=1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from tools.time_pipeline import make
from eryn_amd.ladder import LadderPipeline

T, W, D = map(int, sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
for kind in ("pipe", "single"):
    e = make(T, W, D, (0, T) if kind == "pipe" else None)
    if kind == "pipe":
        LadderPipeline.connect_local([e])
    e.step(100); e.synchronize()
    dt = 1e9
    for _ in range(int(os.environ.get("PIPE_PROF_REPS", "5"))):          # (best of a few blocks: the first one after the warm-up runs cold)
        t0 = time.perf_counter(); e.step(iters); e.synchronize(); dt = min(dt, (time.perf_counter() - t0) / iters * 1e6)
    e.set_profiling(True); e.step(iters); e.synchronize(); tm = e.timing(); e.set_profiling(False)
    s = tm["stretch_ms"] / max(tm["n_stretch"], 1) * 1e3
    f = tm["fused_ms"] / max(tm["n_fused"], 1) * 1e3
    p = tm["pt_ms"] / max(tm["n_pt"], 1) * 1e3
    print(f"{kind:7s} T={T} W={W} D={D} delay={os.environ.get('PIPE_DELAY','0')}: {dt:7.2f} us/iter | stretch launch {s:6.2f} us x{tm['n_stretch']/iters:.0f}, "
          f"fused launch {f:6.2f} us x{tm['n_fused']/iters:.0f}, cascade {p:6.2f} us x{tm['n_pt']/iters:.0f}")
    e.close()
