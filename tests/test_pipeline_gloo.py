"""CPU model of the LADDER PIPELINE's message protocol over gloo, world_size 2 and 3.

The HIP kernels themselves are tested on the GPU (tests/test_hip_pipeline.py); what runs here is the
protocol they implement, with NumPy arithmetic from the oracle and blocking gloo point-to-point messages in
the same order and direction as the mailbox puts:

    LDN   cold -> hot   (L, P) of the cold side's hottest rung after its stretch move (+ its rows: the
                        stand-in for the hot side PULLING rows out of the cold neighbour's pool)
    LUP   hot -> cold   (L, P) of the hot side's coldest rung when its own pairs are done
          both sides evaluate the boundary pair from the same draws
    ROWS  hot -> cold   rows of the walkers that move down (pushed)
    CNT   all -> all    swap counts of the pairs each rank owns (ladder adaptation, replicated)

Every rank must end bit-identical to the unsharded oracle (= the reference) driven by the same draws -
positions, log-likelihood, log-prior, adapted ladder, swap counters - with walkers crossing the boundaries and
falling through a whole rank in one sweep.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eryn_amd.ladder import rung_partition


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _send(a, dst):
    dist.send(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)), dst)


def _recv(shape, src):
    t = torch.empty(shape, dtype=torch.float64)
    dist.recv(t, src)
    return t.numpy()


class PipelineRankModel:
    """One rank of the pipeline: rungs [b, e) of the ladder, rung 0 coldest (tempering.py:515-541 across ranks)."""

    def __init__(self, x, L, P, betas, b, e, rank, nranks, lo, hi, loglike):
        self.x, self.L, self.P = x.copy(), L.copy(), P.copy()
        self.betas = betas.copy()
        self.b, self.e, self.rank, self.nranks = b, e, rank, nranks
        self.T, self.W, self.D = len(betas), x.shape[1], x.shape[2]
        self.lo, self.hi, self.loglike = lo, hi, loglike
        self.time = 0
        self.fell_through = 0

    def stretch(self, draws, orc):
        for sp in (0, 1):
            orc.stretch_split(self.x, self.L, self.P, self.betas[self.b:self.e], draws["labels"], sp, draws[f"rint{sp}"],
                              draws[f"u_zz{sp}"], draws[f"u_acc{sp}"], 2.0, self.lo, self.hi, self.loglike)

    def _decide(self, i, L_hot, L_cold, iperm, i1perm, u):
        dbeta = self.betas[i - 1] - self.betas[i]                      # tempering.py:518-522
        with np.errstate(divide="ignore"):
            return dbeta * (L_hot[iperm] - L_cold[i1perm]) > np.log(u)  # :535-541

    def pt_sweep(self, iperm, i1perm, u_swap, orc):
        """iperm / i1perm / u_swap rows j <-> pair i = T-1-j, as drawn by the reference."""
        T, W, b, e = self.T, self.W, self.b, self.e
        top, bot = e < T, b > 0
        row = lambda i: T - 1 - i                                      # noqa: E731
        counts = np.zeros(T - 1)
        from_top = np.zeros(W, dtype=bool)                             # slots of my hottest rung filled from above this sweep
        if top:                                                        # LDN: what the hot side needs, right after the stretch move
            _send(np.concatenate([self.L[-1], self.P[-1], self.x[-1].ravel()]), self.rank + 1)
        ldn = _recv((2 * W + W * self.D,), self.rank - 1) if bot else None
        if top:
            lup = _recv((2 * W,), self.rank + 1)                       # LUP: (L, P) of rung e when the hot side's pairs are done
            ip, i1p, u = iperm[row(e)], i1perm[row(e)], u_swap[row(e)]
            sel = self._decide(e, lup[:W], self.L[-1], ip, i1p, u)
            counts[e - 1] = sel.sum()                                  # the cold side owns the boundary pair's count
            rows = _recv((int(sel.sum()), self.D), self.rank + 1)      # ROWS: pushed by the hot side
            tl = e - 1 - b
            self.L[tl, i1p[sel]] = lup[:W][ip[sel]]
            self.P[tl, i1p[sel]] = lup[W:][ip[sel]]
            self.x[tl, i1p[sel]] = rows
            from_top[i1p[sel]] = True
        for i in range(e - 1, b, -1):                                  # my own pairs, hot -> cold
            hi_, lo_ = i - b, i - 1 - b
            ip, i1p, u = iperm[row(i)], i1perm[row(i)], u_swap[row(i)]
            sel = self._decide(i, self.L[hi_], self.L[lo_], ip, i1p, u)
            counts[i - 1] = sel.sum()
            a_, c_ = ip[sel], i1p[sel]
            for arr in (self.x, self.L, self.P):
                tmp = arr[hi_, a_].copy()
                arr[hi_, a_] = arr[lo_, c_]
                arr[lo_, c_] = tmp
            if hi_ == e - 1 - b:
                moved = from_top[a_]
                nf = np.zeros(W, dtype=bool)
                nf[c_[moved]] = True
                from_top = nf
            else:
                nf = np.zeros(W, dtype=bool)
                nf[c_[from_top[a_]]] = True
                from_top = nf
        if e - b == 1 and not top:
            from_top[:] = False
        if bot:
            _send(np.concatenate([self.L[0], self.P[0]]), self.rank - 1)                      # LUP
            ip, i1p, u = iperm[row(b)], i1perm[row(b)], u_swap[row(b)]
            Lc, Pc, xc = ldn[:W], ldn[W:2 * W], ldn[2 * W:].reshape(W, self.D)
            sel = self._decide(b, self.L[0], Lc, ip, i1p, u)
            going_down = self.x[0, ip[sel]].copy()
            self.fell_through += int(from_top[ip[sel]].sum()) if (e - b > 1 or top) else 0
            _send(going_down, self.rank - 1)                                                    # ROWS (push)
            self.L[0, ip[sel]] = Lc[i1p[sel]]
            self.P[0, ip[sel]] = Pc[i1p[sel]]
            self.x[0, ip[sel]] = xc[i1p[sel]]                                                   # the pull
        # CNT: every rank learns every pair's count; adaptation replicated (tempering.py:563-596, 632-633)
        t = torch.from_numpy(counts)
        dist.all_reduce(t)
        self.betas = orc.adapt_ladder(self.betas, t.numpy(), W, self.time, 10000, 100)
        self.time += 1
        return t.numpy()


def _worker(rank, world, port, T, W, D, n_iters, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import eryn_oracle as orc
    from tests import parity_utils as pu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o, mu, invcov = pu.make_oracle(T, W, D, box=3.0, x0=np.random.RandomState(1).uniform(-2, 2, size=(T, W, D)))
        _, bounds = rung_partition(T, world)
        b, e = bounds[rank]
        m = PipelineRankModel(o.x[b:e], o.L[b:e], o.P[b:e], o.betas, b, e, rank, world, o.lo, o.hi,
                              lambda x: orc.gaussian_log_like(x, mu, invcov))
        crossed = 0
        for _ in range(n_iters):
            o.iteration()
            rec = o.trace[-1]
            local = dict(labels=rec["labels"][b:e])
            for sp in (0, 1):
                for k in ("rint", "u_zz", "u_acc"):
                    local[f"{k}{sp}"] = rec[f"{k}{sp}"][b:e]
            m.stretch(local, orc)
            swaps = m.pt_sweep(rec["iperm"], rec["i1perm"], rec["u_swap"], orc)
            assert np.array_equal(swaps, rec["swaps_accepted"])
            assert np.array_equal(m.x, rec["x"][b:e])
            assert np.array_equal(m.L, rec["L"][b:e])
            assert np.array_equal(m.P, rec["P"][b:e])
            assert np.array_equal(m.betas, rec["betas_after"])
            if e < T:
                crossed += int(rec["sel"][T - 1 - e].sum())
            o.trace.clear()
        q.put((rank, "ok", crossed, m.fell_through))
    except Exception:                           # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc(), 0, 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T,W,D", [(2, 4, 24, 3), (3, 6, 17, 4), (3, 3, 20, 3)])
def test_pipeline_protocol_matches_unsharded_oracle(world, T, W, D):
    n_iters = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, W, D, n_iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res
    assert sum(r[2] for r in res) > 0, "no walker crossed a shard boundary: the exchange was not exercised"
    if world == 3:
        assert sum(r[3] for r in res) > 0, "no walker fell through a whole rank in one sweep"
