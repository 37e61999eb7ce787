"""CPU pre-validation of candidate cheap permutation primitives (round 3): uniformity of the cascade matching and of the
complement pairs under a Feistel network whose round function is ONE 24-bit multiply, and of block labels that are the top
bit of a small Feistel permutation of the column's place in its block.  Vectorised over keys.  Not product code."""
import sys
import numpy as np

sys.path.insert(0, ".")
from tests import production_draws as pd

U = np.uint64
M24 = U(0xFFFFFF)


def F24(x, k):
    return (((x ^ (k & M24)) * U(0x9E3779)) & U(0xFFFFFFFF)) >> U(12)


def feistel(x, keys, bits, rounds=8, inverse=False):
    """x: (n, m) uint64, keys: list of 8 arrays (n, 1)"""
    lb = bits >> 1
    rb = bits - lb
    lm, rm = U((1 << lb) - 1), U((1 << rb) - 1)
    L, R = x >> U(rb), x & rm
    rs = range(rounds - 2, -1, -2) if inverse else range(0, rounds, 2)
    for r in rs:
        if inverse:
            R = R ^ (F24(L, keys[r + 1]) & rm)
            L = L ^ (F24(R, keys[r]) & lm)
        else:
            L = L ^ (F24(R, keys[r]) & lm)
            R = R ^ (F24(L, keys[r + 1]) & rm)
    return (L << U(rb)) | R


def prp(x, keys, bits, W, rounds=8):
    out = feistel(x, keys, bits, rounds)
    while True:
        bad = out >= U(W)
        if not bad.any():
            return out
        out = np.where(bad, feistel(out, keys, bits, rounds), out)


def chi2_z(counts, expected):
    chi2 = float(((counts - expected) ** 2 / expected).sum())
    dof = counts.size - 1
    return (chi2 - dof) / np.sqrt(2.0 * dof)


def rand_keys(n, seed):
    rs = np.random.RandomState(seed)
    return [rs.randint(0, 2 ** 32, size=(n, 1), dtype=np.uint64) for _ in range(8)]


def test_matching(W, n, rounds):
    bits = pd.idx_bits_of(W)
    x = np.broadcast_to(np.arange(W, dtype=np.uint64), (n, W))
    a = prp(x, rand_keys(n, 1), bits, W, rounds)
    b = prp(x, rand_keys(n, 2), bits, W, rounds)
    assert np.array_equal(np.sort(a, axis=1), x)
    cnt = np.zeros((W, W))
    np.add.at(cnt, (x.astype(np.int64).ravel(), a.astype(np.int64).ravel()), 1)
    z1 = chi2_z(cnt.ravel(), n / W)
    cnt2 = np.zeros((W, W))
    np.add.at(cnt2, (a.astype(np.int64).ravel(), b.astype(np.int64).ravel()), 1)
    z2 = chi2_z(cnt2.ravel(), n / W)
    # second order: (a[c], a[c+1]) joint over W*(W-1) cells
    cnt3 = np.zeros((W, W))
    np.add.at(cnt3, (a[:, :-1].astype(np.int64).ravel(), a[:, 1:].astype(np.int64).ravel()), 1)
    off = cnt3[~np.eye(W, dtype=bool)]
    z3 = chi2_z(off, n * (W - 1) / (W * (W - 1)))
    print(f"  matching W={W} rounds={rounds} keys={n}: z(c->slot) {z1:+.2f}  z(slot_i,slot_i-1) {z2:+.2f}  z(adjacent columns) {z3:+.2f}")


def test_labels(W, cb, n, rounds, top_identity):
    """complement pairs on one rung: column map = prp (or identity on the top rung), rank in block = small Feistel"""
    bits = pd.idx_bits_of(W)
    lcb = cb.bit_length() - 1
    hb, N0 = cb // 2, W // 2
    keys = rand_keys(n, 3)
    c = np.broadcast_to(np.arange(W, dtype=np.uint64), (n, W))
    slot = c if top_identity else prp(c, keys, bits, W, rounds)
    blk = c >> U(lcb)
    rs = np.random.RandomState(9)
    kb = ((blk * U(0x9E3779B1)) ^ keys[0] ^ U(0x7F4A7C15)) & U(0xFFFFFFFF)        # stand-in for fmix32(a0 ^ block)
    kb = pd.fmix32(kb)
    bkeys = [(keys[r] ^ ((kb << U(r)) | (kb >> U(32 - r)))) & U(0xFFFFFFFF) if r else keys[r] ^ kb for r in range(8)]
    rank = feistel(c & U(cb - 1), bkeys, lcb, 8)
    assert np.all(np.sort(rank.reshape(n, W // cb, cb), axis=2) == np.arange(cb))
    h = (rank >= U(hb)).astype(np.int64)
    place = h * N0 + (blk.astype(np.int64)) * hb + (rank.astype(np.int64) % hb)
    order = np.empty((n, W), dtype=np.int64)
    np.put_along_axis(order, place, slot.astype(np.int64), axis=1)
    r = rs.randint(0, N0, size=(n, W))
    s0 = np.arange(W) < N0
    cw = np.take_along_axis(order, np.where(s0, N0, 0) + r, axis=1)
    pair = np.zeros((W, W))
    np.add.at(pair, (order.ravel(), cw.ravel()), 1)
    assert np.all(np.diag(pair) == 0)
    off = pair[~np.eye(W, dtype=bool)]
    z = chi2_z(off, n * W / (W * (W - 1)))
    lab = np.zeros((n, W), dtype=np.int8)
    np.put_along_axis(lab, order[:, N0:], 1, axis=1)
    # pairwise label-difference probability for every walker pair: (W/2)/(W-1)
    diff = (lab[:, :, None] != lab[:, None, :]).mean(axis=0)
    iu = np.triu_indices(W, 1)
    p = (W / 2) / (W - 1)
    zmax = np.abs((diff[iu] - p) / np.sqrt(p * (1 - p) / n)).max()
    print(f"  labels W={W} cb={cb} top_identity={top_identity} keys={n}: z(complement pairs) {z:+.2f}   max |z| of P(labels differ) over pairs {zmax:.2f}")


if False:
    for rounds in (8, 6):
        for W in (64, 100):
            test_matching(W, 40000, rounds)
    test_matching(4096, 300, 8)
    for W, cb in ((64, 64), (64, 8), (64, 4), (64, 2), (96, 32)):
        for top in (False, True):
            test_labels(W, cb, 20000, 8, top)


def _variants():
    global F24
    base = F24

    def F24x2(x, k):
        h = ((x ^ (k & M24)) * U(0x9E3779)) & U(0xFFFFFFFF)
        h ^= h >> U(15)
        h = ((h & M24) * U(0x85EBCB)) & U(0xFFFFFFFF)
        return h >> U(11)

    def Ffmix(x, k):
        return pd.fmix32((x ^ k) & U(0xFFFFFFFF))

    def F32(x, k):                       # one full 32-bit multiply, high bits
        return ((((x ^ k) & U(0xFFFFFFFF)) * U(0x9E3779B1)) & U(0xFFFFFFFF)) >> U(16)

    for name, f, rounds in (("fmix32 x8 (round 2)", Ffmix, 8), ("F24x2 x8", F24x2, 8), ("F24x2 x6", F24x2, 6), ("F32hi x8", F32, 8)):
        F24 = f
        print(name)
        for W in (64, 100):
            test_matching(W, 40000, rounds)
    F24 = base


if __name__ == "__main__":
    _variants()
