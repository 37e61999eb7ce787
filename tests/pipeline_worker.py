"""Worker for tests/test_hip_pipeline.py (run as a subprocess so the HIP runtime sees its env).

  local <nranks> <T> <W> <D> <iters> <out.npz>       N shards in this process on one GPU
  ipc   <rank> <world> <T> <W> <D> <iters> <outdir>  one shard per PROCESS, mailboxes mapped through HIP IPC
  staged <rank> <world> <T> <W> <D> <iters> <outdir> one shard per process, messages by torch.distributed point-to-point
  single <T> <W> <D> <iters> <out.npz>               the unsharded run both are compared with
env PIPE_TEST_DELAY=1 runs everything with adaptation_delay = 1 (then "single" is a 1-rank pipeline).
  timeout                                            rank 1 never steps: rank 0 must raise, not hang
  replay <nranks> <T> <W> <D> <iters>                N local shards against the ORACLE fed with the exported Philox draws
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (before libhipensemble, see eryn_amd/_lib.py)

from eryn_amd.engine import HipEnsemble  # noqa: E402
from eryn_amd.ladder import LadderPipeline, rung_partition  # noqa: E402
from eryn_amd.likelihood import GaussianLikelihood  # noqa: E402
from eryn_amd.moves.tempering import make_ladder  # noqa: E402

SEED = 11


def problem(T, W, D):
    rng = np.random.RandomState(5)
    mu = rng.uniform(-1, 1, size=D)
    a = rng.randn(D, D)
    invcov = a @ a.T / D + np.eye(D)
    x0 = rng.uniform(-3, 3, size=(T, W, D))
    return mu, invcov, x0, make_ladder(D, ntemps=T)


DELAY = int(os.environ.get("PIPE_TEST_DELAY", "0"))      # hens_config.adaptation_delay of every context
MODEL = os.environ.get("PIPE_TEST_MODEL", "gauss")       # "rosen_mix": BASELINE config 4 in small - Rosenbrock + Stretch/Gaussian mix; "gauss_periodic"


def make(T, W, D, rng_range=None, delay=None):
    mu, invcov, x0, betas = problem(T, W, D)
    if MODEL == "rosen_mix":
        from eryn_amd.likelihood import RosenbrockLikelihood
        like = RosenbrockLikelihood(D)
    else:
        like = GaussianLikelihood(mu, invcov)
    e = HipEnsemble(T, W, D, like, -6.0, 6.0, seed=SEED, rung_range=rng_range,
                    adaptation_delay=DELAY if delay is None else delay)
    r0, r1 = rng_range if rng_range else (0, T)
    if MODEL == "gauss_periodic":                        # periodic parameters on every rank - before the pipeline is initialised
        period = np.zeros(D)
        period[1::3] = 5.0
        e.set_periodic(period)
        x0 = x0.copy()
        x0[..., 1::3] = np.mod(x0[..., 1::3], 5.0)
    e.upload(x0[r0:r1], betas=betas)
    e.eval_state()
    if MODEL == "rosen_mix":
        e.set_mh_proposal("iso", 0.02, 0.5)              # half of the iterations are Gaussian MH proposals
    return e


def snapshot(e):
    x, L, P, betas = e.download()
    c = e.counters()
    acc = c["accepted"]
    if MODEL == "rosen_mix":
        m = e.mh_counters()
        assert m["num_proposals"] > 0 and c["num_proposals"] > 0, "the mix must use both moves"
        acc = acc + 1000.0 * m["accepted"]              # both counters in one array (accept counts stay < 1000)
    return dict(x=x, L=L, P=P, betas=betas, accepted=acc, swaps_total=c["swaps_total"], swaps_last=c["swaps_last"])


def main():
    mode = sys.argv[1]
    if mode == "single":
        T, W, D, iters = map(int, sys.argv[2:6])
        e = make(T, W, D)
        if DELAY:                                      # the delayed schedule exists in the pipeline only: one rank of it
            LadderPipeline.connect_local([e])
        for n in (iters // 2, iters - iters // 2):     # two calls: the batch / flush logic at a call boundary
            e.step(n)
        np.savez(sys.argv[6], **snapshot(e))
    elif mode == "local":
        nranks, T, W, D, iters = map(int, sys.argv[2:7])
        _, bounds = rung_partition(T, nranks)
        engs = [make(T, W, D, b) for b in bounds]
        LadderPipeline.connect_local(engs)
        for n in (iters // 2, iters - iters // 2):
            for e in engs:
                e.step(n)                               # asynchronous: the shards run concurrently on the GPU
            for e in engs:
                e.synchronize()
        snaps = [snapshot(e) for e in engs]
        out = {k: np.concatenate([s[k] for s in snaps], axis=0) for k in ("x", "L", "P", "accepted")}
        for k in ("betas", "swaps_total", "swaps_last"):
            for s in snaps[1:]:
                assert np.array_equal(s[k], snaps[0][k]), f"{k} differs between ranks"
            out[k] = snaps[0][k]
        np.savez(sys.argv[7], **out)
    elif mode == "ipc":
        rank, world, T, W, D, iters = map(int, sys.argv[2:8])
        outdir = sys.argv[8]
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)     # only to exchange the handles
        _, bounds = rung_partition(T, world)
        e = make(T, W, D, bounds[rank])
        pipe = LadderPipeline(e, rank, world, dist=dist)
        for n in (iters // 2, iters - iters // 2):
            pipe.step(n)
            e.synchronize()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **snapshot(e))
        dist.barrier()
        e.close()
        dist.destroy_process_group()
    elif mode == "staged":
        # the same protocol with point-to-point messages between the stages (RCCL on a multi-GPU node; gloo here)
        rank, world, T, W, D, iters = map(int, sys.argv[2:8])
        outdir = sys.argv[8]
        import torch.distributed as dist
        from eryn_amd.ladder import StagedPipeline
        dist.init_process_group("gloo", rank=rank, world_size=world)
        _, bounds = rung_partition(T, world)
        e = make(T, W, D, bounds[rank], delay=0)
        pipe = StagedPipeline(e, rank, world, dist, torch.device("cuda", 0))
        for n in (iters // 2, iters - iters // 2):
            pipe.step(n)
            e.synchronize()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **snapshot(e))
        dist.barrier()
        e.close()
        dist.destroy_process_group()
    elif mode == "rccl1":
        # the library's own RCCL transport (hens_comm_init) with ONE rank - what a one-GPU box can run of it: the communicator, a
        # ncclSend / ncclRecv round trip to myself, and hens_step(n) as one call through the staged protocol's three stages
        T, W, D, iters = map(int, sys.argv[2:6])
        from eryn_amd.ladder import RcclPipeline
        e = make(T, W, D, delay=0)
        pipe = RcclPipeline(e, 0, 1)
        v = np.random.RandomState(1).randn(100003)
        back = e.comm_selfsend(v)
        assert np.array_equal(back, v), "ncclSend / ncclRecv to myself did not return the data"
        for n in (iters // 2, iters - iters // 2):
            pipe.step(n)
            e.synchronize()
        np.savez(sys.argv[6], **snapshot(e))
        pipe.close()
        e.close()
    elif mode == "replay":
        # the sharded production path held to the oracle: the draws are a pure function of (seed, iteration, global
        # rung, walker), so a whole-ladder context exports them and the oracle replays the pipeline's iterations
        from oracle import eryn_oracle as orc
        from tests import replay_utils as ru
        nranks, T, W, D, iters = map(int, sys.argv[2:7])
        mu, invcov, x0, betas0 = problem(T, W, D)
        _, bounds = rung_partition(T, nranks)
        engs = [make(T, W, D, b) for b in bounds]
        LadderPipeline.connect_local(engs)
        whole = make(T, W, D)                              # draws + the initial log-likelihoods the device computed
        x, L, P, betas = whole.download()
        st = ru.OracleState(x, L, P, betas)
        fn = lambda q: orc.gaussian_log_like(q, mu, invcov)      # noqa: E731
        done = 0
        for n in (2, iters - 2):
            for e in engs:
                e.step(n)
            for e in engs:
                e.synchronize()
            ru.replay(whole, st, done, n, fn, np.full(D, -6.0), np.full(D, 6.0))
            done += n
            snaps = [e.download() for e in engs]
            cs = [e.counters() for e in engs]
            cnt = dict(accepted=np.concatenate([c["accepted"] for c in cs], axis=0), swaps_total=cs[0]["swaps_total"],
                       swaps_last=cs[0]["swaps_last"])
            ru.assert_state_equal(st, *[np.concatenate([s[k] for s in snaps], axis=0) for k in range(3)], snaps[0][3],
                                  counters=cnt, what=f"{nranks}-shard pipeline after {done} iterations")
        assert st.swaps_total.sum() > 0 and st.min_margin > 1e-12
    elif mode == "timeout":
        # a neighbour that never steps: the waiting rank must fail with an error, not hang the GPU
        _, bounds = rung_partition(4, 2)
        engs = [make(4, 128, 8, b) for b in bounds]
        LadderPipeline.connect_local(engs)
        engs[0].step(2)
        try:
            engs[0].synchronize()
        except RuntimeError as exc:
            print("raised:", exc, flush=True)
        else:
            raise SystemExit("no error although rank 1 never answered")
    print("worker done", mode, flush=True)


if __name__ == "__main__":
    main()
