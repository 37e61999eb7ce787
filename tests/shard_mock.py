"""NumPy stand-in for the HIP shard engine (TEST INFRASTRUCTURE): lets the CPU tests drive
eryn_amd.ladder.ShardedLadder over gloo with world_size 2.  Compute comes from the oracle."""
import numpy as np
import torch

from oracle import eryn_oracle as orc


class NumpyShardEngine:
    def __init__(self, x, L, P, betas, r0, r1, lo, hi, loglike, a=2.0, adaptive=True, lag=10000, nu=100):
        self.x, self.L, self.P = x.copy(), L.copy(), P.copy()          # local rungs [Tl, W, D]
        self.betas = betas.copy()                                      # full ladder
        self.r0, self.r1 = r0, r1
        self.T, self.Tl, self.W, self.D = len(betas), r1 - r0, x.shape[1], x.shape[2]
        self.lo, self.hi, self.loglike, self.a = lo, hi, loglike, a
        self.adaptive, self.lag, self.nu, self.time = adaptive, lag, nu, 0
        self._gather = torch.zeros(self.T, self.W, dtype=torch.float64)
        self.row_doubles = self.D + 2
        self._send = torch.zeros(self.Tl * self.W, self.row_doubles, dtype=torch.float64)
        self._recv = torch.zeros(self.Tl * self.W, self.row_doubles, dtype=torch.float64)

    def stretch(self, draws):
        keeps = []
        for sp in (0, 1):
            out = orc.stretch_split(self.x, self.L, self.P, self.betas[self.r0:self.r1], draws["labels"], sp,
                                    draws[f"rint{sp}"], draws[f"u_zz{sp}"], draws[f"u_acc{sp}"], self.a, self.lo,
                                    self.hi, self.loglike)
            keeps.append(out["keep"])
        return keeps

    def local_logl(self):
        return torch.from_numpy(self.L.copy())

    def gather_buffer(self):
        return self._gather

    def plan(self, rank_of_rung, nranks, rank, draws=None, adapt=True):
        T, W = self.T, self.W
        Lf = self._gather.numpy().copy()
        src = np.arange(T * W).reshape(T, W)
        sel_all = np.zeros((T - 1, W), dtype=bool)
        swaps = np.zeros(T - 1)
        for j, i in enumerate(range(T - 1, 0, -1)):                    # tempering.py:515-559
            ip, i1p = draws["iperm"][j], draws["i1perm"][j]
            dbeta = self.betas[i - 1] - self.betas[i]
            with np.errstate(divide="ignore"):
                sel = dbeta * (Lf[i, ip] - Lf[i - 1, i1p]) > np.log(draws["u_swap"][j])
            sel_all[j] = sel
            swaps[i - 1] = sel.sum()
            a_, b_ = ip[sel], i1p[sel]
            for arr in (Lf, src):
                tmp = arr[i, a_].copy()
                arr[i, a_] = arr[i - 1, b_]
                arr[i - 1, b_] = tmp
        if adapt and self.adaptive:
            self.betas = orc.adapt_ladder(self.betas, swaps, W, self.time, self.lag, self.nu)
            self.time += 1
        # local results + exchange lists
        self._newL = Lf[self.r0:self.r1].copy()
        self._src = src
        oldx, oldP = self.x.copy(), self.P.copy()
        self._newx, self._newP = oldx.copy(), oldP.copy()
        send = np.zeros(nranks, dtype=np.int64)
        recv = np.zeros(nranks, dtype=np.int64)
        rows = {p: [] for p in range(nranks)}
        for t in range(T):
            for w in range(W):
                s = int(src[t, w])
                st, sw = divmod(s, W)
                dr, sr = int(rank_of_rung[t]), int(rank_of_rung[st])
                if dr == rank and sr == rank:
                    self._newx[t - self.r0, w] = oldx[st - self.r0, sw]
                    self._newP[t - self.r0, w] = oldP[st - self.r0, sw]
                elif sr == rank:
                    rows[dr].append(np.concatenate([[np.int64(t * W + w).view(np.float64)], oldx[st - self.r0, sw],
                                                    [oldP[st - self.r0, sw]]]))
                    send[dr] += 1
                elif dr == rank:
                    recv[sr] += 1
        flat = [r for p in range(nranks) for r in rows[p]]
        if flat:
            self._send[:len(flat)] = torch.from_numpy(np.stack(flat))
        return send, recv, sel_all, swaps

    def send_buffer(self, n):
        return self._send[:n]

    def recv_buffer(self, n):
        return self._recv[:n]

    def finish(self, n_recv):
        rec = self._recv[:n_recv].numpy()
        for row in rec:
            g = int(row[:1].view(np.int64)[0])
            t, w = divmod(g, self.W)
            self._newx[t - self.r0, w] = row[1:1 + self.D]
            self._newP[t - self.r0, w] = row[1 + self.D]
        self.x, self.L, self.P = self._newx, self._newL, self._newP
