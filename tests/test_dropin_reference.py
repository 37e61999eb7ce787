"""The HIP-backed move under the REAL reference sampler (build container only; CPU).

``INTEGRATION.md`` section 1 claims ``eryn.ensemble.EnsembleSampler(..., moves=eryn_amd.moves.StretchMove(...))`` is a
drop-in.  Here the unmodified reference (imported from /root/reference/src, never copied) drives
``eryn_amd.moves.StretchMove`` / ``GaussianMove`` through ``run_mcmc``: constructor-time attribute injection
(ensemble.py:517-544), move choice and ``propose`` (:971-984), ``temperature_control.swaps_accepted`` (:977), backend
``save_step``.  The device context is replaced by a stand-in with the same method surface whose compute is the pinned
oracle - this container has no GPU, and what is under test is the plugin protocol, not the kernels (those are pinned by
the -m gpu parity tests).  The chain must equal the fixtures captured from the reference's own StretchMove.

Skipped where /root/reference does not exist (the GPU box)."""
import os
import sys
import types

import numpy as np
import pytest

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")


@pytest.fixture(scope="module")
def eryn():
    for m in ("corner", "seaborn"):            # imported unconditionally by eryn/utils/plot.py
        sys.modules.setdefault(m, types.ModuleType(m))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True             # the reference tree is read-only
    try:
        import eryn.ensemble
        import eryn.prior
    finally:
        sys.dont_write_bytecode = old
    return sys.modules["eryn"]


class OracleEngine:
    """Stand-in for eryn_amd.engine.HipEnsemble: the methods the moves call, computed by the oracle."""

    def __init__(self, T, W, D, loglike, lo, hi, a=2.0, adaptive=True, lag=10000, nu=100, stop=-1):
        from eryn_amd.likelihood import GaussianLikelihood
        self.T, self.W, self.D, self.Tl = T, W, D, T
        self.likelihood = GaussianLikelihood(np.zeros(D), np.ones(D))      # only its type matters here
        self.loglike, self.lo, self.hi, self.a = loglike, np.full(D, lo), np.full(D, hi), a
        self.adaptive, self.lag, self.nu, self.stop, self.time = adaptive, lag, nu, stop, 0
        self.period = None
        self.calls = []

    def set_periodic(self, period):
        self.period = None if period is None else np.array(period, dtype=np.float64)

    def upload(self, x, logl=None, logp=None, betas=None):
        self.x, self.L, self.P = np.array(x, copy=True), np.array(logl, copy=True), np.array(logp, copy=True)
        self.betas = None if betas is None else np.array(betas, copy=True)
        self.calls.append("upload")

    def set_adapt_time(self, t):
        self.time = int(t)

    def stretch_split(self, split, labels, rint, u_zz, u_acc):
        from oracle import eryn_oracle as orc
        out = orc.stretch_split(self.x, self.L, self.P, self.betas, np.asarray(labels), split, rint, u_zz, u_acc, self.a,
                                self.lo, self.hi, self.loglike, period=self.period)
        return out["keep"]

    def mh_step(self, step, u_acc):
        from oracle import eryn_oracle as orc
        return orc.mh_step(self.x, self.L, self.P, self.betas, np.asarray(step), np.asarray(u_acc), self.lo, self.hi,
                           self.loglike, period=self.period)["keep"]

    def pt_sweep(self, iperm, i1perm, u_swap, adapt=True):
        from oracle import eryn_oracle as orc
        sel, sw = orc.pt_sweep(self.x, self.L, self.P, self.betas, iperm, i1perm, u_swap)
        if adapt and self.adaptive:
            if self.stop < 0 or self.time < self.stop:
                self.betas = orc.adapt_ladder(self.betas, sw, self.W, self.time, self.lag, self.nu)
            self.time += 1
        return sel, sw

    def download(self, want_x=True):
        self.calls.append("download")
        return self.x.copy(), self.L.copy(), self.P.copy(), None if self.betas is None else self.betas.copy()

    # -- what the lazy State and the device-draw mode use (round 6) -----------------------------------------------------
    tempered = True

    def download_betas(self):
        return self.betas.copy()

    def set_mh_proposal(self, kind, scale, weight):
        self.calls.append(("mh_proposal", kind, weight))

    def step_report(self, n_iters, n_last=1):
        """Stand-in for hens_step_report: n_iters whole iterations of the pinned oracle on draws of its own (the device draws
        with Philox; what is under test here is the protocol, not the stream)."""
        from oracle import eryn_oracle as orc
        if getattr(self, "_o", None) is None or self._o_src is not self.x:
            self._o = orc.OracleSampler(self.x, self.loglike, self.lo, self.hi, np.random.RandomState(77), np.random.RandomState(78),
                                        betas=self.betas, a=self.a)
            self._o.time = self.time
        o, acc = self._o, np.zeros((self.T, self.W))
        for i in range(n_iters):
            m = o.iteration()
            if i >= n_iters - n_last:
                acc += m
        self.x, self.L, self.P, self.betas, self.time = o.x, o.L, o.P, o.betas, o.time
        self._o_src = self.x
        self.calls.append("step")
        return acc.astype(np.uint8), np.array(o.swaps_accepted, dtype=float), self.betas.copy()


def _fixture(golden_dir, name):
    with np.load(os.path.join(golden_dir, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


def _loglike(x, mu, invcov):                   # tests/test_eryn.py:33-35, batched
    diff = x - mu
    return -0.5 * (diff * np.dot(invcov, diff.T).T).sum(axis=1)


def _reference_sampler(eryn, fx, move, **kw):
    from eryn.ensemble import EnsembleSampler
    from eryn.prior import ProbDistContainer, uniform_dist
    T, W, D, box = int(fx["T"]), int(fx["W"]), int(fx["D"]), float(fx["box"])
    np.random.seed(int(fx["seed_construct"]))  # R := snapshot of the global stream at construction
    priors = ProbDistContainer({i: uniform_dist(-box, box) for i in range(D)})
    if T > 1:
        kw["tempering_kwargs"] = dict(ntemps=T)
    return EnsembleSampler(W, D, _loglike, priors, args=[fx["mu"], fx["invcov"]], vectorize=True, moves=move, **kw)


@pytest.mark.parametrize("name", ["f2_pt", "f3_oddW", "f4_narrowbox"])
def test_hip_stretch_move_under_the_real_sampler_reproduces_the_fixture(eryn, golden_dir, name):
    from eryn_amd.likelihood import GaussianLikelihood
    from eryn_amd.moves import StretchMove
    fx = _fixture(golden_dir, name)
    T, W, D, box, n = int(fx["T"]), int(fx["W"]), int(fx["D"]), float(fx["box"]), int(fx["nsteps"])
    mu, invcov = fx["mu"], fx["invcov"]
    move = StretchMove(a=float(fx["a"]), likelihood=GaussianLikelihood(mu, invcov), prior_box=(-box, box))
    eng = OracleEngine(T, W, D, lambda x: _loglike(x, mu, invcov), -box, box, a=float(fx["a"]))
    move.attach_engine(eng)
    s = _reference_sampler(eryn, fx, move)
    # what the constructor injected (ensemble.py:517-544)
    assert move.temperature_control is s.temperature_control and move.ntemps == T
    assert move.accepted.shape == (T, W) and move.periodic is None
    np.random.seed(int(fx["seed_run"]))
    state = s.run_mcmc(fx["x0"], n, store=True)
    last = f"it{n - 1}_"
    assert np.array_equal(state.branches["model_0"].coords[:, :, 0, :], fx[last + "x"])
    assert np.array_equal(state.log_like, fx[last + "L"]) and np.array_equal(state.log_prior, fx[last + "P"])
    assert np.array_equal(state.betas, fx[last + "betas"])
    assert np.array_equal(s.temperature_control.betas, fx[last + "betas"])
    assert np.array_equal(s.temperature_control.swaps_accepted, fx[last + "swaps_accepted"])
    assert np.array_equal(move.accepted, fx["accepted_total"]) and move.num_proposals == int(fx["num_proposals"])
    # the reference's backend stored the chain the move produced (backend.py:1014-1091)
    chain = s.get_chain()["model_0"]
    assert chain.shape == (n, T, W, 1, D)
    for it in (0, n // 2, n - 1):
        assert np.array_equal(chain[it][:, :, 0, :], fx[f"it{it}_x"])
    assert np.array_equal(s.backend.accepted, fx["accepted_total"])
    assert "upload" in eng.calls


def test_hip_move_mix_under_the_real_sampler(eryn, golden_dir):
    """StretchMove + GaussianMove by weight (ensemble.py:971): the m6 fixture was captured from the reference's own moves"""
    from eryn_amd.likelihood import GaussianLikelihood
    from eryn_amd.moves import GaussianMove, StretchMove
    fx = _fixture(golden_dir, "m6_mix")
    T, W, D, box, n = int(fx["T"]), int(fx["W"]), int(fx["D"]), float(fx["box"]), int(fx["nsteps"])
    mu, invcov = fx["mu"], fx["invcov"]
    like = GaussianLikelihood(mu, invcov)
    eng = OracleEngine(T, W, D, lambda x: _loglike(x, mu, invcov), -box, box)
    assert str(fx["move0_kind"]) == "stretch" and str(fx["move1_kind"]) == "gauss" and str(fx["move1_mode"]) == "vector"
    mvs = [(StretchMove(likelihood=like, prior_box=(-box, box)), float(fx["weights"][0])),
           (GaussianMove({"model_0": float(fx["move1_cov"])}, likelihood=like, prior_box=(-box, box)), float(fx["weights"][1]))]
    for m, _ in mvs:
        m.attach_engine(eng)
    s = _reference_sampler(eryn, fx, mvs)
    np.random.seed(int(fx["seed_run"]))
    state = s.run_mcmc(fx["x0"], n, store=False)
    last = f"it{n - 1}_"
    assert np.array_equal(state.branches["model_0"].coords[:, :, 0, :], fx[last + "x"])
    assert np.array_equal(state.log_like, fx[last + "L"])
    assert np.array_equal(mvs[0][0].accepted, fx["move0_accepted"]) and mvs[0][0].num_proposals == int(fx["move0_num_proposals"])
    assert np.array_equal(mvs[1][0].accepted, fx["move1_accepted"]) and mvs[1][0].num_proposals == int(fx["move1_num_proposals"])


def test_periodic_container_injected_by_the_real_sampler(eryn, golden_dir):
    """The real sampler turns ``periodic={branch: {index: period}}`` into its own PeriodicContainer and assigns it to
    every move that has none (ensemble.py:338-347,528-536).  The device moves read the container's ``inds_periodic`` /
    ``periods`` and hand the periods to their context before every proposal: the chain must equal the p2 fixture,
    captured from the reference's own StretchMove + GaussianMove with the same periodic parameters."""
    from eryn.utils import PeriodicContainer
    from eryn_amd.likelihood import GaussianLikelihood
    from eryn_amd.moves import GaussianMove, StretchMove
    fx = _fixture(golden_dir, "p2_mix_periodic")
    T, W, D, box, n = int(fx["T"]), int(fx["W"]), int(fx["D"]), float(fx["box"]), int(fx["nsteps"])
    mu, invcov = fx["mu"], fx["invcov"]
    like = GaussianLikelihood(mu, invcov)
    eng = OracleEngine(T, W, D, lambda x: _loglike(x, mu, invcov), -box, box)
    mvs = [(StretchMove(likelihood=like, prior_box=(-box, box)), float(fx["weights"][0])),
           (GaussianMove({"model_0": float(fx["move1_cov"])}, likelihood=like, prior_box=(-box, box)), float(fx["weights"][1]))]
    for m, _ in mvs:
        m.attach_engine(eng)
    per = {int(d): float(p) for d, p in enumerate(fx["period"]) if p > 0}
    s = _reference_sampler(eryn, fx, mvs, periodic={"model_0": per})
    assert all(isinstance(m.periodic, PeriodicContainer) for m, _ in mvs)
    np.random.seed(int(fx["seed_run"]))
    state = s.run_mcmc(fx["x0"], n, store=False)
    last = f"it{n - 1}_"
    assert np.array_equal(eng.period, fx["period"])
    assert np.array_equal(state.branches["model_0"].coords[:, :, 0, :], fx[last + "x"])
    assert np.array_equal(state.log_like, fx[last + "L"]) and np.array_equal(state.log_prior, fx[last + "P"])
    assert np.array_equal(state.betas, fx[last + "betas"])
    for i, (m, _) in enumerate(mvs):
        assert np.array_equal(m.accepted, fx[f"move{i}_accepted"]) and m.num_proposals == int(fx[f"move{i}_num_proposals"])


def test_invalid_periodic_injection_fails_loudly(eryn, golden_dir):
    from eryn_amd.likelihood import GaussianLikelihood
    from eryn_amd.moves import StretchMove
    fx = _fixture(golden_dir, "f2_pt")
    box = float(fx["box"])
    move = StretchMove(likelihood=GaussianLikelihood(fx["mu"], fx["invcov"]), prior_box=(-box, box))
    with pytest.raises(ValueError):
        move.periodic = 3.0                    # neither a container nor a dict (ensemble.py:340-345)
    move.periodic = None                       # the no-op assignment stays legal


# ---- round 6: the device-resident State mirror (VERDICT r5 missing #3) -------------------------------------------
def test_lazy_state_under_the_real_sampler_downloads_at_stored_steps_only(eryn, golden_dir):
    """``StretchMove(lazy_state=True)`` hands the unmodified reference sampler a DeviceState: with num_repeats_in_model = 3 and
    thin_by = 2 the reference's loop (ensemble.py:965-1041) proposes 36 times and stores 6 steps - the walkers must cross the
    boundary 6 times (``Backend.save_step`` reads them, backends/backend.py:1049-1090), not 36, and the chain must still be the
    r1 fixture's, captured from the reference's own StretchMove with the same seeds.  A state somebody has read is uploaded
    again before the next proposal (its arrays may have been changed); one nobody has read is not."""
    from eryn_amd.likelihood import GaussianLikelihood
    from eryn_amd.moves import StretchMove
    from eryn_amd.state import DeviceState
    fx = _fixture(golden_dir, "r1_repeats3_thin2")
    T, W, D, box, n = int(fx["T"]), int(fx["W"]), int(fx["D"]), float(fx["box"]), int(fx["nsteps"])
    mu, invcov = fx["mu"], fx["invcov"]
    move = StretchMove(likelihood=GaussianLikelihood(mu, invcov), prior_box=(-box, box), lazy_state=True)
    eng = OracleEngine(T, W, D, lambda x: _loglike(x, mu, invcov), -box, box)
    move.attach_engine(eng)
    fx = dict(fx, a=2.0)
    s = _reference_sampler(eryn, fx, move, num_repeats_in_model=int(fx["repeats"]))
    np.random.seed(int(fx["seed_run"]))
    it = 0
    for state in s.sample(fx["x0"], iterations=n, thin_by=int(fx["thin_by"]), store=True):
        assert isinstance(state, DeviceState) and state.materialized          # (save_step has read it)
        pre = f"it{it}_"
        assert np.array_equal(state.branches["model_0"].coords[:, :, 0, :], fx[pre + "x"])
        assert np.array_equal(state.log_like, fx[pre + "L"]) and np.array_equal(state.log_prior, fx[pre + "P"])
        assert np.array_equal(state.betas, fx[pre + "betas"])
        assert np.array_equal(s.backend.accepted, fx[pre + "backend_accepted"])
        it += 1
    assert np.array_equal(s.get_chain()["model_0"][:, :, :, 0, :], fx["chain"])
    assert np.array_equal(move.accepted, fx["move_accepted"]) and move.num_proposals == int(fx["num_proposals"]) == 36
    assert eng.calls.count("download") == n, f"{eng.calls.count('download')} downloads for {n} stored steps"
    assert eng.calls.count("upload") == n, "one upload at the start + one behind every stored step but the last"


def test_device_draw_move_under_the_real_sampler_thin_by_10(eryn, golden_dir):
    """``StretchMove(rng="philox")``: propose() is ONE device iteration (hens_step_report) and returns the accept mask, the swap
    counts and the ladder - everything the reference's loop reads after a proposal (ensemble.py:974-977) - while the walkers
    stay on the device.  Under the unmodified reference sampler with thin_by = 10, store = True: 40 proposals, 4 stored steps,
    4 downloads, 4 uploads; the backend holds what the device held at the stored steps."""
    from eryn_amd.likelihood import GaussianLikelihood
    from eryn_amd.moves import GaussianMove, StretchMove
    fx = _fixture(golden_dir, "f2_pt")
    T, W, D, box = int(fx["T"]), int(fx["W"]), int(fx["D"]), float(fx["box"])
    mu, invcov = fx["mu"], fx["invcov"]
    like = GaussianLikelihood(mu, invcov)
    move = StretchMove(likelihood=like, prior_box=(-box, box), rng="philox", seed=11)
    assert move.lazy_state
    eng = OracleEngine(T, W, D, lambda x: _loglike(x, mu, invcov), -box, box)
    move.attach_engine(eng)
    s = _reference_sampler(eryn, fx, move)
    seen = []
    for state in s.sample(fx["x0"], iterations=4, thin_by=10, store=True):
        seen.append((eng.x.copy(), eng.L.copy(), eng.betas.copy()))
    assert eng.calls.count("step") == 40 and eng.calls.count("download") == 4 and eng.calls.count("upload") == 4
    chain = s.get_chain()["model_0"]
    assert chain.shape == (4, T, W, 1, D)
    for i, (x, L, b) in enumerate(seen):
        assert np.array_equal(chain[i][:, :, 0, :], x) and np.array_equal(s.get_log_like()[i], L)
        assert np.array_equal(s.get_betas()[i], b)
    assert move.num_proposals == 40 and move.accepted.sum() > 0 and move.accepted.max() <= 40
    assert np.array_equal(s.temperature_control.betas, eng.betas) and s.temperature_control.swaps_accepted.shape == (T - 1,)
    # a state nobody read is stale once the context has stepped on: reading it then fails loudly
    gen = s.sample(state, iterations=2, thin_by=1, store=False)
    first = next(gen)
    next(gen)
    with pytest.raises(RuntimeError):
        first.log_like
    # the Gaussian move of a mix asks the context for its own move before it steps
    g = GaussianMove({"model_0": 0.01}, likelihood=like, prior_box=(-box, box), rng="philox")
    g.attach_engine(eng)
    g.temperature_control = s.temperature_control
    g.accepted = np.zeros((T, W))
    st, acc = g.propose(s.get_model(), state)
    assert ("mh_proposal", "iso", 1.0) in eng.calls and acc.shape == (T, W)


def test_device_draw_move_mix_under_the_real_sampler_keeps_one_resident_state(eryn, golden_dir):
    """Stretch + Gaussian, both rng="philox", under the unmodified reference sampler: the sampler's own stream picks the move per
    proposal (ensemble.py:971); the two moves share one context and one resident state - a DeviceState made by one move is current
    for the other (no upload in between), and every proposal tells the context which move to run."""
    from eryn_amd.likelihood import GaussianLikelihood
    from eryn_amd.moves import GaussianMove, StretchMove
    from eryn_amd.state import DeviceState
    fx = _fixture(golden_dir, "m6_mix")
    T, W, D, box = int(fx["T"]), int(fx["W"]), int(fx["D"]), float(fx["box"])
    mu, invcov = fx["mu"], fx["invcov"]
    like = GaussianLikelihood(mu, invcov)
    eng = OracleEngine(T, W, D, lambda x: _loglike(x, mu, invcov), -box, box)
    shared = [None]
    mvs = [(StretchMove(likelihood=like, prior_box=(-box, box), rng="philox"), 0.5),
           (GaussianMove({"model_0": 0.01}, likelihood=like, prior_box=(-box, box), rng="philox"), 0.5)]
    for m, _ in mvs:
        m.attach_engine(eng, shared)
    s = _reference_sampler(eryn, fx, mvs)
    state = s.run_mcmc(fx["x0"], 30, store=False)
    assert isinstance(state, DeviceState)
    st, g = mvs[0][0], mvs[1][0]
    assert st.num_proposals + g.num_proposals == 30 and st.num_proposals > 3 and g.num_proposals > 3
    assert eng.calls.count("upload") == 1 and eng.calls.count("download") == 0, "one upload at the start, nothing read until the end"
    kinds = [c[1] for c in eng.calls if isinstance(c, tuple) and c[0] == "mh_proposal"]
    assert "iso" in kinds and None in kinds, "the context was told to run the Gaussian move and the stretch move in turn"
    assert np.array_equal(state.log_like, eng.L) and eng.calls.count("download") == 1
