"""GPU test of the sharded-ladder kernels: two contexts on ONE MI355X, each owning half the rungs,
with the all-gather / all-to-all performed by device copies in this process.  (The collective
layer itself is covered on CPU by tests/test_ladder_gloo.py; a single-GPU box cannot host two
RCCL ranks.)  Teacher-forced against the oracle: everything must match the unsharded reference."""
import numpy as np
import pytest

from tests import parity_utils as pu

pytestmark = pytest.mark.gpu


def _make_shards(o, mu, invcov, nranks):
    import torch
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.ladder import HipShardEngine, rung_partition
    from eryn_amd.likelihood import GaussianLikelihood
    rank_of, bounds = rung_partition(o.T, nranks)
    dev = torch.device("cuda", 0)
    shards = []
    for r0, r1 in bounds:
        e = HipEnsemble(o.T, o.W, o.D, GaussianLikelihood(mu, invcov), o.lo, o.hi, a=o.a, rung_range=(r0, r1), seed=7)
        e.upload(o.x[r0:r1], o.L[r0:r1], o.P[r0:r1], o.betas)
        shards.append(HipShardEngine(e, dev, share_stream=False))
    return rank_of, bounds, shards


def _pt_exchange(shards, rank_of, draws, adapt=True):
    import torch
    n = len(shards)
    full = torch.cat([s.local_logl() for s in shards], dim=0)
    for s in shards:
        s.gather_buffer().copy_(full)
    torch.cuda.synchronize()
    plans = [s.plan(rank_of, n, r, draws=draws, adapt=adapt) for r, s in enumerate(shards)]
    moved = 0
    for dst in range(n):
        recv_counts = plans[dst][1]
        ro = 0
        for src in range(n):
            cnt = int(recv_counts[src])
            send_counts = plans[src][0]
            assert int(send_counts[dst]) == cnt, "send/recv counts disagree"
            so = int(send_counts[:dst].sum())
            if cnt:
                shards[dst].recv_buffer(int(recv_counts.sum()))[ro:ro + cnt].copy_(
                    shards[src].send_buffer(int(send_counts.sum()))[so:so + cnt])
            ro += cnt
            moved += cnt
    torch.cuda.synchronize()
    for r, s in enumerate(shards):
        s.finish(int(plans[r][1].sum()))
    return plans, moved


@pytest.mark.parametrize("T,W,D,nranks", [(4, 128, 8, 2), (8, 256, 32, 4), (6, 70, 5, 2)])
def test_sharded_teacher_forced(T, W, D, nranks):
    o, mu, invcov = pu.make_oracle(T, W, D, box=3.0, x0=np.random.RandomState(1).uniform(-2, 2, size=(T, W, D)))
    rank_of, bounds, shards = _make_shards(o, mu, invcov, nranks)
    total_moved = 0
    for _ in range(4):
        prev = (o.x.copy(), o.L.copy(), o.P.copy(), o.betas.copy(), o.time)
        o.iteration()
        rec = o.trace[-1]
        for (r0, r1), s in zip(bounds, shards):
            s.e.upload(prev[0][r0:r1], prev[1][r0:r1], prev[2][r0:r1], prev[3])
            s.e.set_adapt_time(prev[4])
            local = dict(labels=rec["labels"][r0:r1])
            for sp in (0, 1):
                for k in ("rint", "u_zz", "u_acc"):
                    local[f"{k}{sp}"] = rec[f"{k}{sp}"][r0:r1]
            keeps = s.stretch(local)
            for sp in (0, 1):
                assert np.array_equal(keeps[sp], rec[f"keep{sp}"][r0:r1])
            x, L, P, _ = s.e.download()
            assert np.array_equal(x, rec["x_after1"][r0:r1])
            # teacher-force the oracle's log-likelihoods so the swap test sees identical bits
            s.e.upload(rec["x_after1"][r0:r1], rec["L_stretch"][r0:r1], rec["P_stretch"][r0:r1], prev[3])
            s.e.set_adapt_time(prev[4])
        draws = {k: rec[k] for k in ("iperm", "i1perm", "u_swap")}
        plans, moved = _pt_exchange(shards, rank_of, draws)
        total_moved += moved
        for (r0, r1), s, pl in zip(bounds, shards, plans):
            assert np.array_equal(pl[2], rec["sel"])
            assert np.array_equal(pl[3], rec["swaps_accepted"])
            x, L, P, betas = s.e.download()
            assert np.array_equal(x, rec["x"][r0:r1]), "rows after the sharded PT exchange"
            assert np.array_equal(L, rec["L"][r0:r1])
            assert np.array_equal(P, rec["P"][r0:r1])
            np.testing.assert_allclose(betas, rec["betas_after"], rtol=1e-13, atol=0)
        o.trace.clear()
    assert total_moved > 0
    for s in shards:
        s.e.close()


def test_sharded_philox_matches_single_context():
    """Production draws: a 2-shard ladder must walk exactly the chain of one whole-ladder context
    (Philox draws depend only on seed, iteration, global rung and walker)."""
    import torch
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import GaussianLikelihood
    T, W, D = 4, 256, 16
    o, mu, invcov = pu.make_oracle(T, W, D, box=50.0, record=False)
    whole = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), o.lo, o.hi, seed=7)
    whole.upload(o.x, o.L, o.P, o.betas)
    rank_of, bounds, shards = _make_shards(o, mu, invcov, 2)
    for it in range(5):
        whole.step(1)
        for s in shards:
            s.stretch()
        _pt_exchange(shards, rank_of, None)
        xw, Lw, Pw, bw = whole.download()
        for (r0, r1), s in zip(bounds, shards):
            x, L, P, b = s.e.download()
            assert np.array_equal(x, xw[r0:r1]), f"iteration {it}"
            assert np.array_equal(L, Lw[r0:r1]) and np.array_equal(P, Pw[r0:r1])
            assert np.array_equal(b, bw)
    whole.close()
    for s in shards:
        s.e.close()
