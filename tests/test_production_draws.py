"""The production RNG construction, pinned.

CPU: the NumPy specification (tests/production_draws.py) reproduces the published known-answer vectors of
Philox4x32-10 (Random123 kat_vectors), its Feistel permutations are bijections with a working inverse, and the plan it
derives has the structure the reference guarantees (red_blue.py:119-124, 150-197).
GPU (-m gpu): ``hens_debug_draws`` - the draws the timed path consumes - equals the specification bit for bit.
"""
import numpy as np
import pytest

from tests import production_draws as pd

KAT = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
       ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
       ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox4x32_10_known_answers(ctr, key, want):
    got = pd.philox4x32_10(*ctr, *key)
    assert tuple(int(v) for v in got) == want


def test_philox_is_vectorised_consistently():
    c2 = np.arange(7, dtype=np.uint64) * 1234567
    a = pd.philox4x32_10(3, 0, c2, 9, 11, 12)
    for i, v in enumerate(c2):
        b = pd.philox4x32_10(3, 0, int(v), 9, 11, 12)
        assert [int(x[i]) for x in a] == [int(x) for x in b]


@pytest.mark.parametrize("W", [2, 3, 33, 64, 100, 257, 4096])
def test_feistel_permutation_is_a_bijection_with_inverse(W):
    key = pd.prp_key(77, 5, pd.PURPOSE_PTPERM, 3)
    bits, x = pd.idx_bits_of(W), np.arange(W)
    y = pd.prp(x, key, bits, W)
    assert np.array_equal(np.sort(y), x)
    assert np.array_equal(pd.prp(y, key, bits, W, inverse=True), x)


@pytest.mark.parametrize("T,W", [(16, 256), (8, 64), (64, 128), (2, 128), (3, 100), (5, 33), (10, 256), (3, 96), (20, 64),
                                 (48, 66)])
def test_plan_structure(T, W):
    cb = pd.label_cb(T, W)
    N0 = (W + 1) // 2
    d = pd.plan(2024, 11, T, W, cb)
    bits = pd.idx_bits_of(W)
    for t in range(T):
        own, cw = d["own"][t], d["cw"][t]
        assert np.array_equal(np.sort(own), np.arange(W))
        first = set(own[:N0].tolist())
        assert all(int(c) not in first for c in cw[:N0]) and all(int(c) in first for c in cw[N0:])
        if cb:                                            # cb/2 walkers of each half per block and rung, listed block by block
            key = pd.prp_key(2024, 11, pd.PURPOSE_PTPERM, t)
            for h, half in enumerate((own[:N0], own[N0:])):
                col = pd.prp(half, key, bits, W, inverse=True)
                assert np.array_equal(col // cb, np.arange(N0) // (cb // 2))
                assert np.all((col % cb >= cb // 2) == bool(h))
        else:
            assert np.all(np.diff(own[:N0]) > 0) and np.all(np.diff(own[N0:]) > 0)     # ascending halves
    assert np.all((d["u_zz"] >= 0) & (d["u_zz"] < 1) & (d["u_acc"] >= 0) & (d["u_acc"] < 1))


@pytest.mark.gpu
@pytest.mark.parametrize("T,W,D,its", [(16, 256, 32, (0, 5)), (64, 128, 8, (3,)), (2, 128, 16, (1,)), (3, 100, 8, (0, 4)),
                                       (8, 4096, 16, ((1 << 32) + 3,)), (10, 256, 32, (2,)), (3, 96, 16, (0,)),
                                       (24, 64, 8, (7,))])
def test_device_draws_equal_the_specification(T, W, D, its):
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import GaussianLikelihood
    from oracle import eryn_oracle as orc
    seed = 0x1234567890ABCDEF
    eng = HipEnsemble(T, W, D, GaussianLikelihood(np.zeros(D), np.ones(D)), -50.0, 50.0, seed=seed)
    eng.upload(np.random.RandomState(0).randn(T, W, D), betas=orc.make_ladder(D, ntemps=T))
    eng.eval_state()
    cb = pd.label_cb(T, W)
    for it in its:
        dev = eng.debug_draws(it)
        spec = pd.plan(seed, it, T, W, cb)
        for k in ("own", "cw"):
            assert np.array_equal(dev[k].astype(np.int64), spec[k]), f"{k} at iteration {it}"
        for k in ("u_zz", "u_acc"):
            assert np.array_equal(dev[k], spec[k]), f"{k} at iteration {it}"
        slot, u = pd.pt_draws(seed, it, T, W)
        assert np.array_equal(dev["pt_slot"].astype(np.int64), slot)
        assert np.array_equal(dev["u_swap"], u)
    eng.close()
