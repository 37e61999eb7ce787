"""Round 6: sweep of the first-round start offset of a CU's second workgroup (hens_kernels.h: stagger_start) on the AQL path.
    python tools/stagger_sweep.py T W D like "0,16,32,..." [launch_mask] [reps]
One process, a fresh context per setting (the environment is read when a context is created), `reps` alternations; the timing is
bench.py's: blocks of 20 steps, median of 100 blocks."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from eryn_amd.engine import HipEnsemble  # noqa: E402
from eryn_amd.likelihood import GaussianLikelihood, RosenbrockLikelihood  # noqa: E402
from eryn_amd.moves.tempering import make_ladder  # noqa: E402

T, W, D, like = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
settings = [int(v) for v in sys.argv[5].split(",")]
mask = sys.argv[6] if len(sys.argv) > 6 else "3"
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 2
steps = 20
bench.BLOCKS = 100


def run(st):
    os.environ["HENS_STAGGER"] = str(st)
    os.environ["HENS_STAGGER_LAUNCH"] = mask
    if like == "rosen":
        eng = HipEnsemble(T, W, D, RosenbrockLikelihood(D), -5.0, 5.0, seed=2024)
        x0 = np.clip(1.0 + 0.05 * np.random.RandomState(1).randn(T, W, D), -4.9, 4.9)
    else:
        mu, invcov = bench.gaussian_problem(D)
        eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, np.diag(invcov).copy() if like == "diag" else invcov), -50.0, 50.0, seed=2024)
        x0 = np.random.RandomState(1).randn(T, W, D)
    eng.upload(x0, betas=make_ladder(D, ntemps=T))
    eng.eval_state()
    eng.step(100)
    eng.synchronize()
    times, _ = bench.timed_blocks(eng.step, eng.synchronize, steps)
    tm = bench.profiled_pass(eng, steps, calls=5)
    x, L, P, b = eng.download()
    eng.close()
    h = float(np.sum(x[0, :8]) + L.sum())
    return float(np.median(times)) / steps * 1e6, tm["stretch_ms"] / max(tm["n_stretch"], 1) * 1e3, tm["fused_ms"] / max(tm["n_fused"], 1) * 1e3, h


for r in range(reps):
    for st in settings:
        us, k1, k2, h = run(st)
        print(f"{T}x{W}x{D} {like} mask {mask} stagger {st & 0xFFFF:5d}{' (dispatch order)' if st & (1 << 30) else ''}: {us:7.2f} us/iter   launch 1 {k1:6.2f}  launch 2 {k2:6.2f}   state hash {h:.10e}", flush=True)
