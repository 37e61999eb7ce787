"""Round 6: k_stretch2 (hens_tile2.h: the persistent, software-pipelined first launch) against k_stretch_fast's rounds of workgroups.
    python tools/tile2_check.py T W D like [iters]
Runs the shape in two subprocesses - default and HENS_NO_TILE2=1 (HENS_TILE2_FORCE=1 in the first one if the grid is small) - and
compares the final state bit for bit; prints bench.py-style timings (blocks of 20, median of 100) and per-launch durations."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import os, sys, hashlib
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
import bench
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood, RosenbrockLikelihood
from eryn_amd.moves.tempering import make_ladder
T, W, D, like, iters = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6])
bench.BLOCKS = 100
if like == "rosen":
    eng = HipEnsemble(T, W, D, RosenbrockLikelihood(D), -5.0, 5.0, seed=2024)
    x0 = np.clip(1.0 + 0.05 * np.random.RandomState(1).randn(T, W, D), -4.9, 4.9)
else:
    mu, invcov = bench.gaussian_problem(D)
    eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, np.diag(invcov).copy() if like == "diag" else invcov), -50.0, 50.0, seed=2024)
    x0 = np.random.RandomState(1).randn(T, W, D)
eng.upload(x0, betas=make_ladder(D, ntemps=T))
eng.eval_state()
eng.step(3); eng.step(iters)
x, L, P, b = eng.download()
c = eng.counters()
h = hashlib.sha256(b"".join(np.ascontiguousarray(v).tobytes() for v in (x, L, P, b, c["accepted"], c["swaps_total"]))).hexdigest()[:16]
times, _ = bench.timed_blocks(eng.step, eng.synchronize, 20)
tm = bench.profiled_pass(eng, 20, calls=5)
print("RESULT", h, float(np.median(times)) / 20 * 1e6, tm["stretch_ms"] / max(tm["n_stretch"], 1) * 1e3, tm["fused_ms"] / max(tm["n_fused"], 1) * 1e3, flush=True)
eng.close()
"""


def run(T, W, D, like, iters, env):
    e = dict(os.environ, **env)
    for k in ("HENS_NO_TILE2", "HENS_TILE2_FORCE") if "HENS_KEEP_ENV" not in os.environ else ():
        if k not in env:
            e.pop(k, None)
    r = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(T), str(W), str(D), like, str(iters)], env=e, capture_output=True, text=True, timeout=600)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT"):
            _, h, us, k1, k2 = line.split()
            return h, float(us), float(k1), float(k2)
    raise RuntimeError(r.stdout[-2000:] + r.stderr[-2000:])


if __name__ == "__main__":
    T, W, D, like = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 300
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    for _ in range(reps):
        a = run(T, W, D, like, iters, {"HENS_TILE2_FORCE": "1"})
        b = run(T, W, D, like, iters, {"HENS_NO_TILE2": "1"})
        print(f"{T}x{W}x{D} {like}: tile2 {a[1]:7.2f} us/iter (launch 1 {a[2]:6.2f}, launch 2 {a[3]:6.2f}) | rounds {b[1]:7.2f} us/iter (launch 1 {b[2]:6.2f}, "
              f"launch 2 {b[3]:6.2f}) | state {'IDENTICAL' if a[0] == b[0] else 'DIFFERS ' + a[0] + ' ' + b[0]}", flush=True)
