"""Where does a sharded-ladder iteration spend its time? (world size 1, collectives forced)"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, ".")
from eryn_amd.engine import HipEnsemble
from eryn_amd.ladder import HipShardEngine, ShardedLadder
from eryn_amd.likelihood import GaussianLikelihood
from eryn_amd.moves.tempering import make_ladder

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
share = int(os.environ.get("SHARE", "1"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
T, W, D = 8, 16384, 64
rs = np.random.RandomState(0)
A = rs.randn(D, D)
mu = 0.1 * rs.randn(D)
invcov = np.linalg.inv(A @ A.T / D + np.eye(D))
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=1)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=make_ladder(D, ntemps=T))
eng.eval_state()
sh = HipShardEngine(eng, dev, share_stream=bool(share))
lad = ShardedLadder(sh, T, dist=dist, rank=0, nranks=1)
lad.force_collectives = True
acc = {}


def tic(name, fn):
    torch.cuda.synchronize(); eng.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize(); eng.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return out


N = 100
for it in range(N + 10):
    if it == 10:
        acc.clear()
    tic("stretch", sh.stretch)
    local = tic("local_logl", sh.local_logl)
    full = sh.gather_buffer()
    tic("all_gather", lambda: dist.all_gather_into_tensor(full.view(-1), local.reshape(-1)))
    send, recv, sel, swaps = tic("plan", lambda: sh.plan(lad.rank_of_rung, 1, 0))
    tic("all_to_all", lambda: dist.all_to_all_single(sh.recv_buffer(int(recv.sum())), sh.send_buffer(int(send.sum())),
                                                      output_split_sizes=[int(recv.sum())], input_split_sizes=[int(send.sum())]))
    tic("finish", lambda: sh.finish(int(recv.sum())))
print("share_stream", share, {k: round(v / N * 1e6, 1) for k, v in acc.items()}, "us per iteration")
t0 = time.perf_counter()
lad.step(N)
eng.synchronize(); torch.cuda.synchronize()
print("lad.step:", (time.perf_counter() - t0) / N * 1e6, "us/iter")
# host-side time of each call, no syncs in between
acc.clear()
def htic(name, fn):
    t0 = time.perf_counter(); out = fn(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return out
t0 = time.perf_counter()
for it in range(N):
    htic("stretch", sh.stretch)
    local = htic("local_logl", sh.local_logl)
    full = sh.gather_buffer()
    htic("all_gather", lambda: dist.all_gather_into_tensor(full.view(-1), local.reshape(-1)))
    send, recv, sel, swaps = htic("plan", lambda: sh.plan(lad.rank_of_rung, 1, 0))
    ns, nr = int(send.sum()), int(recv.sum())
    htic("all_to_all", lambda: dist.all_to_all_single(sh.recv_buffer(nr), sh.send_buffer(ns), output_split_sizes=[nr], input_split_sizes=[ns]))
    htic("finish", lambda: sh.finish(nr))
eng.synchronize(); torch.cuda.synchronize()
print("unsynced loop:", (time.perf_counter() - t0) / N * 1e6, "us/iter; host time per call", {k: round(v / N * 1e6, 1) for k, v in acc.items()})
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); lad.step(N); eng.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
dist.destroy_process_group()
