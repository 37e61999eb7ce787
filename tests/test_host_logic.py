"""CPU tests: host-side mirror of the reference interface and the C-ABI surface (no GPU needed)."""
import os
import re

import numpy as np
import pytest

from eryn_amd import _build, _lib
from eryn_amd.moves.tempering import TemperatureControl, make_ladder
from eryn_amd.prior import ProbDistContainer, uniform_dist
from eryn_amd.state import State

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    _build.build()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "hipensemble.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(hens_[a-z_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), f"libhipensemble.so does not export {name}"
    assert b"gfx950" in lib.hens_version()


def test_fence_free_kernels_store_census_is_the_reviewed_one():
    """The stepping launches of one GPU carry no release fence on their AQL packet (hens_kernels.h: wt_store, launch_end_wait): every
    store a later launch reads must be write-through, and one plain store added to k_stretch_fast / k_stretch2 / k_split1_pt / k_iter
    would be a silent stale read on another XCD.  The ISA cannot say which stores a later launch reads, but it can say that their
    number has not changed since somebody looked: per fence-free kernel, the counts of global stores, of plain ones (no sc0 / sc1 / nt)
    and of atomics in the built library against the reviewed census (tools/store_census.py --write after the review)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import store_census
    _build.build()
    got = store_census.census()
    want = json.load(open(store_census.GOLDEN))["kernels"]
    assert sorted(got) == sorted(want), ("fence-free instantiations changed: " +
                                         ", ".join(sorted(set(got) ^ set(want))[:6]) + " - review and run tools/store_census.py --write")
    moved = {k: (want[k], got[k]) for k in got if got[k] != want[k]}
    assert not moved, (f"{len(moved)} fence-free kernels changed their store counts, e.g. {list(moved.items())[:3]}: is every store a later "
                       f"launch reads still write-through?  Then python tools/store_census.py --write")
    assert len(got) >= 100 and all(v["stores"] > v["plain"] for v in got.values())


def test_struct_layout_matches_header():
    import ctypes as C
    # hens_config: 12 x i32, i64, 4 x f64, u64
    assert C.sizeof(_lib.HensConfig) == 12 * 4 + 8 + 4 * 8 + 8
    assert C.sizeof(_lib.HensTiming) == 4 * 8 + 4 * 8 + 8 + 8 + 8      # + fused_ms, n_fused, clock


def test_no_gpu_fails_loudly():
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import GaussianLikelihood
    lib = _lib.load()
    if lib.hens_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        HipEnsemble(1, 16, 2, GaussianLikelihood(np.zeros(2), np.eye(2)), -1, 1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "eryn_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"


def test_make_ladder_matches_reference_fixture(golden_dir):
    fx = np.load(os.path.join(golden_dir, "ladders.npz"))
    for key in fx.files:
        parts = key.split("_")
        D = int(parts[0][1:])
        if key.endswith("_inf"):
            got = make_ladder(D, ntemps=int(parts[1][1:]), Tmax=np.inf)
        elif parts[1].startswith("Tmax"):
            got = make_ladder(D, Tmax=float(parts[1][4:]))
        elif len(parts) == 3:
            got = make_ladder(D, ntemps=int(parts[1][1:]), Tmax=float(parts[2][4:]))
        else:
            got = make_ladder(D, ntemps=int(parts[1][1:]))
        assert np.array_equal(got, fx[key]), key


def test_make_ladder_errors():
    with pytest.raises(ValueError):
        make_ladder(0, ntemps=3)
    with pytest.raises(ValueError):
        make_ladder(3)
    with pytest.raises(ValueError):
        make_ladder(3, Tmax=0.5)
    with pytest.raises(ValueError):
        make_ladder(3, ntemps=2.5)


def test_temperature_control_defaults_and_draw_order(golden_dir):
    fx = np.load(os.path.join(golden_dir, "f2_pt.npz"))
    T, W, D = int(fx["T"]), int(fx["W"]), int(fx["D"])
    tc = TemperatureControl(D, W, ntemps=T)
    assert np.array_equal(tc.betas, fx["betas0"])
    assert tc.adaptive and tc.permute and tc.adaptation_lag == 10000 and tc.adaptation_time == 100
    assert np.array_equal(tc.swaps_proposed, np.full(T - 1, W))
    # draw order on the global stream: T label shuffles, then (perm, perm, uniform) per pair hot -> cold
    np.random.seed(int(fx["seed_run"]))
    labels = np.tile(np.arange(W), (T, 1)) % 2
    [np.random.shuffle(r) for r in labels]
    assert np.array_equal(labels, fx["it0_labels"])
    ip, i1p, u = tc.draw_swap_randoms()
    assert np.array_equal(ip, fx["it0_iperm"]) and np.array_equal(i1p, fx["it0_i1perm"])
    assert np.array_equal(u, fx["it0_u_swap"])


def test_tempered_posterior_helper():
    tc = TemperatureControl(3, 8, betas=np.array([1.0, 0.5, 0.0]))
    logl = np.array([[-1.0, -np.inf], [-2.0, -1e300], [-3.0, -np.inf]])
    logp = np.zeros_like(logl)
    out = tc.compute_log_posterior_tempered(logl, logp)
    assert out[0, 0] == -1.0 and out[1, 0] == -1.0 and out[2, 0] == 0.0
    assert out[2, 1] == -np.inf          # 0 * -inf = NaN -> -inf (tempering.py:343-349)


def test_state_reshapes_like_reference():
    x = np.zeros((3, 8, 4))
    s = State(x, log_like=np.zeros((3, 8)), log_prior=np.zeros((3, 8)), betas=np.ones(3))
    assert s.branches["model_0"].shape == (3, 8, 1, 4)
    assert s.branches["model_0"].inds.all() and s.branches["model_0"].inds.shape == (3, 8, 1)
    s2 = State(np.zeros((8, 4)))
    assert s2.branches["model_0"].shape == (1, 8, 1, 4)
    s3 = State(s, copy=True)
    s3.branches["model_0"].coords[0, 0, 0, 0] = 5.0
    assert s.branches["model_0"].coords[0, 0, 0, 0] == 0.0
    with pytest.raises(ValueError):
        State(np.zeros(4))
    with pytest.raises(ValueError):
        State([1, 2, 3])
    assert s.get_log_posterior(temper=True).shape == (3, 8)


def test_prior_container_box():
    pri = ProbDistContainer({0: uniform_dist(-5, 5), 1: uniform_dist(3, 1)})
    lo, hi = pri.box_bounds()
    assert list(lo) == [-5, 1] and list(hi) == [5, 3]
    x = pri.rvs(size=(4, 10))
    assert x.shape == (4, 10, 2) and (x[..., 1] >= 1).all() and (x[..., 1] <= 3).all()
    with pytest.raises(ValueError):
        uniform_dist(1, 1)
    from eryn_amd.engine import box_logp_inside
    assert box_logp_inside(lo, hi) == np.log(1 / 10.0) + np.log(1 / 2.0)


def test_sampler_callable_needs_numpy_rng_and_a_gpu():
    from eryn_amd.ensemble import EnsembleSampler
    pri = {0: uniform_dist(-1, 1), 1: uniform_dist(-1, 1)}
    with pytest.raises(NotImplementedError, match="host-callable"):
        EnsembleSampler(16, 2, lambda x: 0.0, pri, rng="philox")
    with pytest.raises(ValueError):
        EnsembleSampler(16, 2, 3.0, pri)
    if _lib.load().hens_device_count() == 0:
        with pytest.raises(RuntimeError, match="no HIP device"):      # no CPU fallback for the device half
            EnsembleSampler(16, 2, lambda x: 0.0, pri)


def test_host_likelihood_contract():
    """eryn_amd.likelihood.HostLikelihood.evaluate mirrors compute_log_like (ensemble.py:1219-1545)."""
    from eryn_amd.likelihood import HostLikelihood
    calls = []

    def f(x, shift):
        calls.append(x.shape)
        return -0.5 * ((x - shift) ** 2).sum(axis=1)

    hl = HostLikelihood(f, 3, args=[0.5])
    q = np.arange(24, dtype=float).reshape(2, 4, 3) / 10
    inbox = np.array([[1, 0, 1, 1], [0, 0, 1, 1]], dtype=bool)
    ll = hl.evaluate(q, inbox)
    assert calls == [(5, 3)]                                  # only in-prior walkers are evaluated, flattened C order
    assert np.all(ll[~inbox] == -1e300)
    assert np.array_equal(ll[inbox], -0.5 * ((q[inbox] - 0.5) ** 2).sum(axis=1))
    with pytest.warns(UserWarning):
        assert np.all(hl.evaluate(q, np.zeros((2, 4), bool)) == -1e300)
    with pytest.raises(ValueError, match="Nan"):
        HostLikelihood(lambda x: np.full(len(x), np.nan), 3).evaluate(q, inbox)
    with pytest.raises(ValueError, match="infinite"):
        hl.evaluate(np.full((1, 2, 3), np.inf), np.ones((1, 2), bool))
    per_walker = HostLikelihood(lambda x, s: -0.5 * ((x - s) ** 2).sum(), 3, args=[0.5], vectorize=False)
    assert np.array_equal(per_walker.evaluate(q, inbox)[inbox], ll[inbox])


@pytest.mark.parametrize("cov,mode,factor", [(0.05, "vector", None), (0.2, "random", 2.0), (0.15, "sequential", None),
                                             ("full", "vector", None), ("full", "vector", 1.5)])
def test_gaussian_move_draws_match_the_pinned_restatement(cov, mode, factor):
    """eryn_amd.moves.GaussianMove.get_step consumes the sampler's RandomState exactly like the oracle's
    GaussianProposal, which tests/test_oracle_golden.py pins bit-for-bit on fixtures captured from the reference
    (gaussian.py:161-176, 265-268) - so the host mirror's proposal steps are the reference's, draw for draw."""
    from eryn_amd.moves import GaussianMove
    from oracle import eryn_oracle as orc
    D, n = 4, 37
    if cov == "full":
        a = np.random.RandomState(3).randn(D, D)
        cov = 0.05 * (a @ a.T / D + np.eye(D))
    mv = GaussianMove({"model_0": cov}, mode=mode, factor=factor)
    pr = orc.GaussianProposal(cov, mode=mode, factor=factor)
    r1, r2 = np.random.RandomState(11), np.random.RandomState(11)
    for _ in range(5):                                  # several calls: the sequential mode's index advances
        assert np.array_equal(mv.get_step(r1, n, D), pr.draw_step(r2, n, D))
    assert np.array_equal(r1.rand(3), r2.rand(3))       # both streams ended in the same state


def test_rj_records_pack_and_unpack_roundtrip():
    """Leaf-packing records (eryn_amd.rj): branches + inds -> records -> branches, leaf masks as integers, the reference's
    NaN fill of unused leaves on snapshots (backends/backend.py:1049-1059) - host logic only, no device."""
    from eryn_amd.rj import RJEngine, TemplateBranch
    eng = RJEngine.__new__(RJEngine)                       # packing needs no context
    eng.branches = [TemplateBranch("gauss", "pulse", [(2.5, 3.5), (-1, 1), (0.01, 0.21)], 10, 0),
                    TemplateBranch("sine", "sine", [(0.5, 1.5), (1, 20), (0, 2 * np.pi)], 4, 1)]
    eng.T, eng.W = 3, 5
    eng.ncoord = 10 * 3 + 4 * 3
    eng.RW = eng.ncoord + 2
    eng.off = np.array([0, 30])
    rs = np.random.RandomState(0)
    x = {"gauss": rs.randn(3, 5, 10, 3), "sine": rs.randn(3, 5, 4, 3)}
    inds = {"gauss": rs.rand(3, 5, 10) < 0.4, "sine": rs.rand(3, 5, 4) < 0.6}
    rec = eng.pack(x, inds)
    assert rec.shape == (3, 5, 44) and eng.RW % 2 == 0
    assert np.array_equal(rec[:, :, 42], (inds["gauss"] * (1 << np.arange(10))).sum(-1))
    x2, inds2 = eng.unpack(rec)
    for k in x:
        assert np.array_equal(x2[k], x[k]) and np.array_equal(inds2[k], inds[k])
    x3, _ = eng.unpack(rec, nan_fill=True)
    assert np.isnan(x3["gauss"][~inds["gauss"]]).all() and np.array_equal(x3["sine"][inds["sine"]], x["sine"][inds["sine"]])
    assert np.array_equal(eng.pack(x3, inds), np.where(np.isnan(eng.pack(x3, inds)), 0, eng.pack(x3, inds)))   # NaN never reaches a record
    # leaf prior accumulated like the reference's container (prior.py:364-383)
    b = eng.branches[0]
    acc = np.zeros(1)
    for d in range(3):
        acc += np.log(1 / (b.hi[d] - b.lo[d]))
    assert b.leaf_logp == acc[0]
    steps = {"gauss": np.ones((3, 5, 10, 3)), "sine": 2 * np.ones((3, 5, 4, 3))}
    sr = eng.steps_to_records(steps)
    assert sr.shape == (3, 5, 42) and np.all(sr[:, :, :30] == 1) and np.all(sr[:, :, 30:] == 2)


def test_padded_row_width_policy(monkeypatch):
    """Rows of Gaussian-likelihood contexts are padded to the next compile-time kernel width; Rosenbrock (coupled sum),
    host-callable likelihoods (proposals travel unpadded) and widths above 128 are not (eryn_amd/engine.py)."""
    from eryn_amd import _lib
    from eryn_amd.engine import padded_width

    class L:
        def __init__(self, kind):
            self.kind = kind
    monkeypatch.delenv("HENS_NO_PAD", raising=False)
    g, gd = L(_lib.LIKE_GAUSS_DENSE), L(_lib.LIKE_GAUSS_DIAG)
    assert [padded_width(d, g) for d in (1, 5, 8, 9, 16, 17, 33, 64, 65, 128, 129, 200)] == \
        [8, 8, 8, 16, 16, 32, 64, 64, 128, 128, 129, 200]
    assert padded_width(11, gd) == 16
    assert padded_width(11, L(_lib.LIKE_ROSENBROCK)) == 11 and padded_width(11, L(_lib.LIKE_HOST)) == 11
    monkeypatch.setenv("HENS_NO_PAD", "1")
    assert padded_width(11, g) == 11
