"""world_size-2 test of the ladder-sharding protocol (eryn_amd/ladder.py) over gloo on CPU.

Each rank owns half the rungs, runs the stretch step locally, all-gathers the log-likelihoods,
replays the whole swap cascade, exchanges only the walker rows that change rank, and must end
bit-identical to the unsharded oracle (= the reference) driven by the same draws.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eryn_amd.ladder import ShardedLadder, rung_partition


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, W, D, n_iters, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests import parity_utils as pu
    from tests.shard_mock import NumpyShardEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o, mu, invcov = pu.make_oracle(T, W, D, box=3.0, x0=np.random.RandomState(1).uniform(-2, 2, size=(T, W, D)))
        _, bounds = rung_partition(T, world)
        r0, r1 = bounds[rank]
        from oracle import eryn_oracle as orc
        eng = NumpyShardEngine(o.x[r0:r1], o.L[r0:r1], o.P[r0:r1], o.betas, r0, r1, o.lo, o.hi,
                               lambda x: orc.gaussian_log_like(x, mu, invcov))
        lad = ShardedLadder(eng, T, dist=dist, rank=rank, nranks=world)
        moved = 0
        for _ in range(n_iters):
            o.iteration()
            rec = o.trace[-1]
            draws = {k: rec[k] for k in ("iperm", "i1perm", "u_swap")}
            local = dict(labels=rec["labels"][r0:r1])
            for sp in (0, 1):
                for k in ("rint", "u_zz", "u_acc"):
                    local[f"{k}{sp}"] = rec[f"{k}{sp}"][r0:r1]
            keeps = eng.stretch(local)
            for sp in (0, 1):
                assert np.array_equal(keeps[sp], rec[f"keep{sp}"][r0:r1])
            sel, swaps = lad.pt_step(draws=draws)
            assert np.array_equal(sel, rec["sel"])
            assert np.array_equal(swaps, rec["swaps_accepted"])
            assert np.array_equal(eng.x, rec["x"][r0:r1])
            assert np.array_equal(eng.L, rec["L"][r0:r1])
            assert np.array_equal(eng.P, rec["P"][r0:r1])
            assert np.array_equal(eng.betas, rec["betas_after"])
            moved += int((eng._src[r0:r1] // W < r0).sum() + (eng._src[r0:r1] // W >= r1).sum())
            o.trace.clear()
        q.put((rank, "ok", moved))
    except Exception as e:                      # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc(), 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("T,W,D", [(4, 24, 3), (6, 17, 4)])
def test_sharded_ladder_matches_unsharded_oracle(T, W, D):
    world, n_iters = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, W, D, n_iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res
    assert sum(r[2] for r in res) > 0, "no walker crossed the shard boundary: the exchange was not exercised"


def test_rung_partition():
    rk, b = rung_partition(8, 4)
    assert list(rk) == [0, 0, 1, 1, 2, 2, 3, 3] and b == [(0, 2), (2, 4), (4, 6), (6, 8)]
    with pytest.raises(ValueError):
        rung_partition(6, 4)


# ---- LadderPipeline construction is failure-atomic across ranks (world_size 2, gloo) --------------------------------
class _FakeEngine:
    """The three calls LadderPipeline.__init__ makes on a shard engine; `fail` makes this rank's step raise."""

    def __init__(self, fail):
        self.fail = fail

    def pipe_init(self, nranks, rank):
        if self.fail == "init":
            raise RuntimeError("hipIpcGetMemHandle refused")
        return bytes([rank]) * 128

    def pipe_connect(self, handles):
        assert len(handles) == 2 * 128
        if self.fail == "connect":
            raise RuntimeError("hipIpcOpenMemHandle refused")


def _pipe_worker(rank, world, port, fail_rank, fail, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from eryn_amd.ladder import LadderPipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        try:
            LadderPipeline(_FakeEngine(fail if rank == fail_rank else None), rank, world, dist=dist, selftest=False)
            out = "ok"
        except RuntimeError as exc:
            out = f"raised: {exc}"
        # every rank is still in step with the others: a further collective completes
        t = torch.ones(1)
        dist.all_reduce(t)
        q.put((rank, out, float(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail", [None, "init", "connect"])
def test_ladder_pipeline_constructor_is_failure_atomic(fail):
    """One rank's pipe_init / pipe_connect fails: EVERY rank must raise (nobody is left inside a collective its peer
    never enters), and the process group must still be usable afterwards (the bench's fallback relies on it)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, 1, fail, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, total in res:
        assert total == world
        if fail is None:
            assert out == "ok"
        else:
            assert out.startswith("raised") and "ranks [1]" in out, out
