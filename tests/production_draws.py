"""NumPy specification of the PRODUCTION random draws of ``hens_step`` (test infrastructure).

Not a restatement of the reference (that is ``oracle/``): the reference draws from NumPy's global Mersenne Twister,
which a device cannot reproduce in parallel.  Production stepping draws from Philox4x32-10 (Salmon et al. 2011), a
pure function of (seed, iteration, purpose, global rung, walker), and builds its permutations from a keyed Feistel
network.  This file states that construction independently of the HIP code (eryn_amd/csrc/hens_kernels.h:
philox4x32_10, u01, stretch_draw, prp_key, pmix, prp, place_column, k_plan, k_plan_draws, stretch_draws_at, pt_slot,
pt_uniform) so that

* the generator can be pinned on CPU against the published known-answer vectors of Random123 (test_host_logic), and
* ``hens_debug_draws`` - hence the draws the timed path consumes - can be compared with it bit for bit on the GPU.

Distributional equivalence with the reference's draws (balanced random halves, uniform complement, uniform matchings,
uniform accept numbers) is what tests/test_hip_rng.py measures; tests/replay_utils.py turns these draws into the
reference's own variables and replays them through the oracle.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
PURPOSE_STRETCH, PURPOSE_STRETCH_ACC, PURPOSE_SPLIT, PURPOSE_PTPERM, PURPOSE_PTU = 0, 2, 8, 9, 10


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over the counter words (arrays or scalars of any broadcastable shape); returns four uint32 arrays."""
    c = [np.asarray(v, dtype=np.uint64) & MASK for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & MASK, p1 & MASK,
             ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in c]


def u01(hi, lo):
    """53-bit uniform in [0, 1) from two words."""
    v = (np.asarray(hi, dtype=np.uint64) << np.uint64(32)) | np.asarray(lo, dtype=np.uint64)
    return (v >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def fmix32(h):
    h = np.asarray(h, dtype=np.uint64) & MASK
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & MASK
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & MASK
    h ^= h >> np.uint64(16)
    return h


def prp_key(seed, it, purpose, rung):
    lo, hi = seed & 0xFFFFFFFF, seed >> 32
    a = philox4x32_10(it & 0xFFFFFFFF, it >> 32, rung, purpose, lo, hi)
    b = philox4x32_10(it & 0xFFFFFFFF, it >> 32, rung, purpose ^ 0x100, lo, hi)
    return [int(v) for v in a + b]


def idx_bits_of(W):
    b, n = 0, 1
    while n < W:
        n <<= 1
        b += 1
    return b


def pmix(x, k):
    """Round function of the Feistel network: two 24-bit multiplies with a shift-xor between them (hens_kernels.h: pmix)."""
    m24 = np.uint64(0xFFFFFF)
    h = (((np.asarray(x, dtype=np.uint64) ^ np.uint64(k)) & m24) * np.uint64(0x9E3779)) & MASK
    h ^= h >> np.uint64(15)
    h = ((h & m24) * np.uint64(0x85EBCB)) & MASK
    return h >> np.uint64(11)


def _feistel(x, key, bits, inverse):
    lb = bits >> 1
    rb = bits - lb
    lm, rm = np.uint64((1 << lb) - 1), np.uint64((1 << rb) - 1)
    L, R = x >> np.uint64(rb), x & rm
    rounds = range(6, -1, -2) if inverse else range(0, 8, 2)
    for r in rounds:
        if inverse:
            R = R ^ (pmix(L, key[r + 1]) & rm)
            L = L ^ (pmix(R, key[r]) & lm)
        else:
            L = L ^ (pmix(R, key[r]) & lm)
            R = R ^ (pmix(L, key[r + 1]) & rm)
    return (L << np.uint64(rb)) | R


def prp(x, key, bits, W, inverse=False):
    """Keyed permutation of [0, W): 8-round alternating Feistel network on ``bits`` bits (round function pmix),
    cycle-walked into range."""
    x = np.array(x, dtype=np.uint64, copy=True)
    out = _feistel(x, key, bits, inverse)
    while True:
        bad = out >= np.uint64(W)
        if not bad.any():
            return out.astype(np.int64)
        out[bad] = _feistel(out[bad], key, bits, inverse)


def place_column(h, p, cb):
    """Column that meets the walker at place p of half h: block * cb + h * cb/2 + member (hens_kernels.h: place_column)."""
    hb = cb // 2
    return (p // hb) * cb + h * hb + (p % hb)


def label_cb(T, W, tempered=True):
    """Columns per block of the block-balanced labelling, or 0 (hens_create)."""
    if tempered and 2 <= T <= 64:
        cb = 1
        while cb * 2 * T <= 128:          # the largest power of two with cb T <= 128 (= 128 / T where T divides 128)
            cb *= 2
        if cb >= 2 and W % cb == 0:
            return cb
    return 0


def stretch_draw(seed, it, wid):
    """(u_zz, u_acc, r22) of split position ``wid`` = rung * W + position: ONE Philox call (stretch_draw in
    hens_kernels.h) - two 53-bit uniforms and, from the 2 x 11 low bits they do not use, a 22-bit number that indexes the
    complement half."""
    lo, hi = seed & 0xFFFFFFFF, seed >> 32
    d = philox4x32_10(it & 0xFFFFFFFF, it >> 32, wid, PURPOSE_STRETCH, lo, hi)
    r22 = ((d[1].astype(np.uint64) & np.uint64(0x7FF)) << np.uint64(11)) | (d[3].astype(np.uint64) & np.uint64(0x7FF))
    return u01(d[0], d[1]), u01(d[2], d[3]), r22


def plan(seed, it, T, W, cb):
    """own, cw, u_zz, u_acc of iteration ``it`` for the whole ladder, [T][W] by split position (k_plan; with block-balanced
    labels k_plan_draws / stretch_draws_at).  Without block labels both halves are listed in ascending walker order; with
    them the walker at place p of half h is the one the column place_column(h, p) meets."""
    bits, N0 = idx_bits_of(W), (W + 1) // 2
    own = np.empty((T, W), dtype=np.int64)
    cw = np.empty((T, W), dtype=np.int64)
    uz = np.empty((T, W))
    ua = np.empty((T, W))
    w = np.arange(W)
    s0 = w < N0
    for t in range(T):
        key = prp_key(seed, it, PURPOSE_PTPERM if cb else PURPOSE_SPLIT, t)
        if cb:
            order = prp(place_column((~s0).astype(np.int64), np.where(s0, w, w - N0), cb), key, bits, W)
        else:
            lab = (prp(w, key, bits, W) >= N0).astype(np.int64)
            order = np.concatenate([w[lab == 0], w[lab == 1]])                   # both halves ascending
            assert (lab == 0).sum() == N0
        own[t] = order
        uz[t], ua[t], r22 = stretch_draw(seed, it, t * W + w)
        Nc = np.where(s0, W - N0, N0).astype(np.uint64)
        r = ((r22 * Nc) >> np.uint64(22)).astype(np.int64)                       # uniform index into the other half
        cw[t] = order[np.where(s0, N0, 0) + r]
    return dict(own=own, cw=cw, u_zz=uz, u_acc=ua)


def pt_draws(seed, it, T, W):
    """pt_slot [T][W] (column c meets slot pt_slot[t, c]) and u_swap [T-1][W] (row j: pair T-1-j)."""
    bits = idx_bits_of(W)
    c = np.arange(W)
    slot = np.empty((T, W), dtype=np.int64)
    for t in range(T):
        slot[t] = prp(c, prp_key(seed, it, PURPOSE_PTPERM, t), bits, W)
    lo, hi = seed & 0xFFFFFFFF, seed >> 32
    u = np.empty((T - 1, W))
    for j in range(T - 1):
        d = philox4x32_10(it & 0xFFFFFFFF, it >> 32, j * W + c, PURPOSE_PTU, lo, hi)
        u[j] = u01(d[0], d[1])
    return slot, u
