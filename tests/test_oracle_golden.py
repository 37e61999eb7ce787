"""Pin oracle/eryn_oracle.py against fixtures captured from the real reference.

Every fixture under tests/golden/ was produced by tests/golden/make_golden.py,
which imports mikekatz04/Eryn itself.  The oracle regenerates both random
streams from the two seeds (R: construction-time snapshot, G: run-time global)
and must reproduce every draw, mask, index, counter and float bit-for-bit.
"""
import os

import numpy as np
import pytest

from oracle import eryn_oracle as orc

FIXTURES = ["f1_plumbing", "f2_pt", "f3_oddW", "f4_narrowbox", "f5_noadapt",
            "f5_nopermute", "f6_medium", "f7_tmaxinf", "f8_nsplits3"]


def _loglike(fx):
    mu, invcov = fx["mu"], fx["invcov"]
    if bool(fx["vectorize"]):
        return lambda x: orc.gaussian_log_like(x, mu, invcov)
    # non-vectorised reference path: one call per walker (ensemble.py:1471-1481)
    return lambda x: np.array([-0.5 * ((xi - mu) * np.dot(invcov, (xi - mu).T).T).sum() for xi in x])


def build_oracle(fx, record=True):
    T, W, D = int(fx["T"]), int(fx["W"]), int(fx["D"])
    R = np.random.RandomState(int(fx["seed_construct"]))
    G = np.random.RandomState(int(fx["seed_run"]))
    box = float(fx["box"])
    kw = {}
    if "betas0" in fx.files:
        kw.update(betas=fx["betas0"], adaptive=bool(fx["adaptive"]), permute=bool(fx["permute"]))
    if "nsplits" in fx.files:
        kw["nsplits"] = int(fx["nsplits"])
    return orc.OracleSampler(fx["x0"], _loglike(fx), np.full(D, -box), np.full(D, box), R, G,
                             a=float(fx["a"]), record=record, **kw)


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype.kind in "biu" or b.dtype.kind in "biu":
        assert np.array_equal(a, b), what
    else:
        # bit-for-bit, NaN-aware
        assert np.array_equal(a, b, equal_nan=True), (what, np.nanmax(np.abs(a - b)))


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_reproduces_reference(name, golden_dir):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    T = int(fx["T"])
    o = build_oracle(fx)
    _same(o.P, fx["P0"], "P0")
    _same(o.L, fx["L0"], "L0")
    if "betas0" in fx.files:
        _same(o.betas, fx["betas0"], "betas0")
    for it in range(int(fx["nsteps"])):
        o.iteration()
        rec, pre = o.trace[-1], f"it{it}_"
        _same(rec["labels"], fx[pre + "labels"], "labels")
        for sp in range(o.nsplits):
            for k in ("rint", "u_zz", "u_acc", "factors", "logp", "logl", "keep"):
                _same(rec[f"{k}{sp}"], fx[pre + f"{k}{sp}"], f"{pre}{k}{sp}")
            if pre + f"q{sp}" in fx.files:
                _same(rec[f"q{sp}"], fx[pre + f"q{sp}"], f"{pre}q{sp}")
            S, _ = orc.split_index_lists(rec["labels"], sp, o.nsplits)
            _same(S, fx[pre + f"S{sp}"], f"{pre}S{sp}")
        if pre + "sel" in fx.files:
            _same(rec["L_stretch"], fx[pre + "L_stretch"], "L_stretch")
            _same(rec["P_stretch"], fx[pre + "P_stretch"], "P_stretch")
            if pre + "iperm" in fx.files:
                _same(rec["iperm"], fx[pre + "iperm"], "iperm")
                _same(rec["i1perm"], fx[pre + "i1perm"], "i1perm")
            _same(rec["u_swap"], fx[pre + "u_swap"], "u_swap")
            _same(rec["sel"], fx[pre + "sel"], "sel")
            _same(rec["swaps_accepted"], fx[pre + "swaps_accepted"], "swaps_accepted")
            _same(rec["betas_after"], fx[pre + "betas"], "betas")
        _same(rec["x"], fx[pre + "x"], pre + "x")
        _same(rec["L"], fx[pre + "L"], pre + "L")
        _same(rec["P"], fx[pre + "P"], pre + "P")
    _same(o.accepted, fx["accepted_total"], "accepted_total")
    assert o.num_proposals == int(fx["num_proposals"])


def test_make_ladder_matches_reference(golden_dir):
    fx = np.load(os.path.join(golden_dir, "ladders.npz"))
    for key in fx.files:
        parts = key.split("_")
        D = int(parts[0][1:])
        if key.endswith("_inf"):
            got = orc.make_ladder(D, ntemps=int(parts[1][1:]), Tmax=np.inf)
        elif parts[1].startswith("Tmax"):
            got = orc.make_ladder(D, Tmax=float(parts[1][4:]))
        elif len(parts) == 3:
            got = orc.make_ladder(D, ntemps=int(parts[1][1:]), Tmax=float(parts[2][4:]))
        else:
            got = orc.make_ladder(D, ntemps=int(parts[1][1:]))
        _same(got, fx[key], key)


def test_fill_path_exercised(golden_dir):
    """The narrow-box fixture must actually hit the -inf prior / -1e300 fill path."""
    fx = np.load(os.path.join(golden_dir, "f4_narrowbox.npz"))
    n_inf = sum(int(np.isinf(fx[f"it{i}_logp{sp}"]).sum()) for i in range(int(fx["nsteps"])) for sp in (0, 1))
    n_fill = sum(int((fx[f"it{i}_logl{sp}"] == -1e300).sum()) for i in range(int(fx["nsteps"])) for sp in (0, 1))
    assert n_inf > 50 and n_fill == n_inf


# ---- Metropolis-Hastings moves: fixtures from tests/golden/make_golden_mh.py -----------------
MH_FIXTURES = ["m1_gauss_iso", "m3_gauss_full", "m4_gauss_random_factor", "m5_gauss_sequential", "m6_mix",
               "m7_gauss_untempered", "m8_mix_narrowbox",
               # periodic parameters (utils/periodic.py through stretch.py:136-154 / gaussian.py:110-115)
               "p1_stretch_periodic", "p2_mix_periodic", "p3_stretch_periodic_untempered"]


def mh_moves_from_fixture(fx, make_gauss):
    moves = []
    for i in range(int(fx["nmoves"])):
        if str(fx[f"move{i}_kind"]) == "stretch":
            moves.append(("stretch", float(fx["weights"][i])))
        else:
            factor = float(fx[f"move{i}_factor"])
            moves.append((make_gauss(fx[f"move{i}_cov"], str(fx[f"move{i}_mode"]), None if np.isnan(factor) else factor),
                          float(fx["weights"][i])))
    return moves


def build_mh_oracle(fx, record=True):
    D = int(fx["D"])
    R = np.random.RandomState(int(fx["seed_construct"]))
    G = np.random.RandomState(int(fx["seed_run"]))
    box = float(fx["box"])
    mu, invcov = fx["mu"], fx["invcov"]
    kw = {}
    if "betas0" in fx.files:
        kw.update(betas=fx["betas0"])
    if "period" in fx.files:
        kw.update(period=fx["period"])
    moves = mh_moves_from_fixture(fx, lambda cov, mode, factor: orc.GaussianProposal(cov, mode=mode, factor=factor))
    return orc.OracleSampler(fx["x0"], lambda x: orc.gaussian_log_like(x, mu, invcov), np.full(D, -box), np.full(D, box),
                             R, G, record=record, moves=moves, **kw)


@pytest.mark.parametrize("name", MH_FIXTURES)
def test_oracle_reproduces_reference_mh(name, golden_dir):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    o = build_mh_oracle(fx)
    picked = np.zeros(int(fx["nmoves"]), dtype=int)
    for it in range(int(fx["nsteps"])):
        o.iteration()
        rec, pre = o.trace[-1], f"it{it}_"
        assert rec["move"] == int(fx[pre + "move"]), "move choice (ensemble.py:971)"
        picked[rec["move"]] += 1
        if "mh_q" in rec:
            _same(rec["mh_q"], fx[pre + "mh_q"], pre + "q")
            _same(rec["mh_logp"], fx[pre + "mh_logp"], pre + "logp")
            _same(rec["mh_logl"], fx[pre + "mh_logl"], pre + "logl")
            _same(rec["mh_keep"], fx[pre + "mh_keep"], pre + "keep")
            # the accept uniforms are the last R draw of the proposal
            kinds = list(fx[pre + "r_kinds"])
            assert kinds[-1] == "rand"
            _same(rec["mh_u_acc"], fx[pre + f"r{len(kinds) - 1}"], pre + "u_acc")
        _same(rec["x"], fx[pre + "x"], pre + "x")
        _same(rec["L"], fx[pre + "L"], pre + "L")
        _same(rec["P"], fx[pre + "P"], pre + "P")
        if o.tempered:
            _same(o.betas, fx[pre + "betas"], pre + "betas")
            _same(o.swaps_accepted, fx[pre + "swaps_accepted"], pre + "swaps_accepted")
    for i in range(int(fx["nmoves"])):
        assert o.move_num_proposals[i] == int(fx[f"move{i}_num_proposals"]) == picked[i]
        _same(o.move_accepted[i], fx[f"move{i}_accepted"], f"move{i}.accepted")


def test_periodic_fixtures_take_both_branches_of_the_distance(golden_dir):
    """The periodic fixtures must wrap proposals and measure distances the short way round (periodic.py:96-112): the
    chains live in [0, period) on the periodic parameters after the first accepted move, and the oracle's distance
    differs from the plain difference somewhere."""
    fx = np.load(os.path.join(golden_dir, "p1_stretch_periodic.npz"))
    per = fx["period"]
    idx = np.flatnonzero(per > 0)
    assert idx.size == 2
    x_last = fx[f"it{int(fx['nsteps']) - 1}_x"]
    inside = lambda x: ((x[..., idx] >= 0.0) & (x[..., idx] < per[idx])).mean()      # noqa: E731
    # (a walker that never accepted a move keeps its unwrapped start, and swaps carry such walkers between rungs)
    assert inside(fx["x0"]) < 0.6 and inside(x_last) > 0.9
    rs = np.random.RandomState(0)
    scale = np.where(per > 0, per, 2.0)                # inside one period: a single shift reaches the short way round
    s, c = rs.uniform(0, 1, size=(50, 4)) * scale, rs.uniform(0, 1, size=(50, 4)) * scale
    d = orc.periodic_distance(s, c, per)
    assert np.array_equal(d[:, per == 0], (c - s)[:, per == 0])
    assert np.all(np.abs(d[:, idx]) <= per[idx] / 2.0 + 1e-12) and not np.array_equal(d, c - s)
    q = orc.periodic_wrap(np.array([[-0.25, 7.0, -3.0, 1.0]]), per)
    assert np.array_equal(q, np.array([[-0.25 % per[0], 7.0, -3.0 % 1.5, 1.0]]))


def test_oracle_reproduces_reference_sampler_loop_with_repeats_and_thinning(golden_dir):
    """ensemble.py:963-1045 with num_repeats_in_model = 3 and thin_by = 2 (tests/golden/make_golden_repeats.py): one stored
    step = thin_by x repeats proposals, each with its own move choice from R; the backend's accept mask is the LAST thinned
    sub-iteration's, summed over its repeats, the stored swap counts the last repeat's."""
    fx = np.load(os.path.join(golden_dir, "r1_repeats3_thin2.npz"))
    T, W, D = int(fx["T"]), int(fx["W"]), int(fx["D"])
    reps, thin, box = int(fx["repeats"]), int(fx["thin_by"]), float(fx["box"])
    mu, invcov = fx["mu"], fx["invcov"]
    o = orc.OracleSampler(fx["x0"], lambda x: orc.gaussian_log_like(x, mu, invcov), np.full(D, -box), np.full(D, box),
                          np.random.RandomState(int(fx["seed_construct"])), np.random.RandomState(int(fx["seed_run"])),
                          betas=orc.make_ladder(D, ntemps=T))
    backend_acc, backend_sw = np.zeros((T, W)), np.zeros(T - 1)
    for it in range(int(fx["nsteps"])):
        for _ in range(thin):
            accepted = np.zeros((T, W))
            for _ in range(reps):
                accepted += o.iteration()
        backend_acc += accepted
        backend_sw += o.swaps_accepted
        pre = f"it{it}_"
        _same(o.x, fx[pre + "x"], pre + "x")
        _same(o.L, fx[pre + "L"], pre + "L")
        _same(o.P, fx[pre + "P"], pre + "P")
        _same(o.betas, fx[pre + "betas"], pre + "betas")
        _same(o.swaps_accepted, fx[pre + "swaps_accepted"], pre + "swaps")
        _same(backend_acc, fx[pre + "backend_accepted"], pre + "backend accepted")
        _same(backend_sw, fx[pre + "backend_swaps"], pre + "backend swaps")
    _same(o.accepted, fx["move_accepted"], "move.accepted")
    assert o.num_proposals == int(fx["num_proposals"]) == int(fx["nsteps"]) * thin * reps
