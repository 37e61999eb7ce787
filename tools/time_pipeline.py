"""Protocol overhead of the ladder pipeline on ONE GPU (several shards share the device, so this is not a
scaling number: it bounds what the flags / puts / extra launches cost per iteration).

  python tools/time_pipeline.py single T W D iters
  python tools/time_pipeline.py local nranks T W D iters        (N contexts in this process)
  python tools/time_pipeline.py ipc world T W D iters           (spawns `world` processes)
  python tools/time_pipeline.py _ipc rank world T W D iters     (internal)
"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from eryn_amd.engine import HipEnsemble  # noqa: E402
from eryn_amd.ladder import LadderPipeline, rung_partition  # noqa: E402
from eryn_amd.likelihood import GaussianLikelihood  # noqa: E402
from eryn_amd.moves.tempering import make_ladder  # noqa: E402


def make(T, W, D, rr=None):
    rng = np.random.RandomState(5)
    mu = rng.uniform(-1, 1, size=D)
    a = rng.randn(D, D)
    invcov = a @ a.T / D + np.eye(D)
    r0, r1 = rr if rr else (0, T)
    e = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=3, rung_range=rr,
                    adaptive=os.environ.get("PIPE_ADAPTIVE", "1") != "0",
                    adaptation_delay=int(os.environ.get("PIPE_DELAY", "0")) if rr is not None else 0)
    e.upload(np.random.RandomState(1).randn(T, W, D)[r0:r1], betas=make_ladder(D, ntemps=T))
    e.eval_state()
    return e


def timed(engs, iters, reps=5):
    best = 1e9
    for _ in range(reps):
        for e in engs:
            e.synchronize()
        t0 = time.perf_counter()
        chunk = iters if len(engs) == 1 else 25      # in-process shards: keep every stream's queue shallow
        for _ in range(iters // chunk):
            for e in engs:
                e.step(chunk)
        for e in engs:
            e.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / iters * 1e6


def main():
    mode = sys.argv[1]
    if mode == "single":
        T, W, D, iters = map(int, sys.argv[2:6])
        e = make(T, W, D)
        e.step(200)
        print(f"single            T={T} W={W} D={D}: {timed([e], iters):8.2f} us/iter")
    elif mode == "local":
        n, T, W, D, iters = map(int, sys.argv[2:7])
        _, bounds = rung_partition(T, n)
        engs = [make(T, W, D, b) for b in bounds]
        LadderPipeline.connect_local(engs)
        for e in engs:
            e.step(200)
        print(f"local x{n} shards  T={T} W={W} D={D}: {timed(engs, iters):8.2f} us/iter (shards share one GPU)")
        if os.environ.get("HENS_PIPE_STATS"):
            for r, e in enumerate(engs):
                st = e.pipe_debug_stats()
                print(f"   rank {r} mean wait per workgroup [us]:", {k: round(v[0] / max(v[1], 1) * 1e6, 2) for k, v in st.items()})
    elif mode == "ipc":
        world = int(sys.argv[2])
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 1000))
        ps = [subprocess.Popen([sys.executable, __file__, "_ipc", str(r)] + sys.argv[2:], env=env) for r in range(world)]
        for p in ps:
            p.wait(timeout=600)
    elif mode == "_ipc":
        rank, world, T, W, D, iters = map(int, sys.argv[2:8])
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        _, bounds = rung_partition(T, world)
        e = make(T, W, D, bounds[rank])
        pipe = LadderPipeline(e, rank, world, dist=dist)
        pipe.step(200)
        e.synchronize()
        best = 1e9
        for _ in range(5):
            e.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            pipe.step(iters)
            e.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"ipc rank {rank}/{world}     T={T} W={W} D={D}: {best / iters * 1e6:8.2f} us/iter (processes share one GPU)", flush=True)
        if os.environ.get("HENS_PIPE_STATS"):
            st = e.pipe_debug_stats()
            n_it = 200 + 5 * iters
            print(f"   rank {rank} mean wait per workgroup [us]:", {k: round(v[0] / max(v[1], 1) * 1e6, 2) for k, v in st.items()},
                  "waits/iter:", {k: round(v[1] / n_it, 1) for k, v in st.items()}, flush=True)
        dist.barrier()
        e.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
