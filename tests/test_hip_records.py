"""The record mode of ``hens_step`` (-m gpu).

Inside a ``hens_step`` call the two-launch iteration keeps {L, P, row} in one 32-byte record per walker and updates
rows in place; every other entry point works on the by-field arrays and the two-home copying scheme.  The boundary
between the two must be invisible:

* splitting a run into calls changes nothing (pack / unpack at the ends of every call),
* the by-field arrays are whole after a call: the teacher-forced parity API continues from that state and agrees with
  the oracle, then ``hens_step`` continues from the parity API's state (copying launches flip the pool half, in-place
  iterations must leave the free half free),
* the three-launch path (``HENS_NO_FUSED=1``: copying half-steps + stand-alone cascade on the same draws) reaches the
  same state bit for bit, with and without the Metropolis-Hastings move in the mix,
* so does the one-launch iteration of small shapes (``k_iter``: versioned rows, replayed complements) against the two
  launches (``HENS_NO_ITER=1``).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import eryn_oracle as orc
from tests import parity_utils as pu
from tests import replay_utils as ru

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _engine(T, W, D, seed=5, mh=None, like_kind="dense"):
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import GaussianLikelihood, RosenbrockLikelihood
    mu, invcov = pu.gaussian_problem(D)
    like = GaussianLikelihood(mu, invcov) if like_kind == "dense" else RosenbrockLikelihood(D)
    box = 50.0 if like_kind == "dense" else 5.0
    eng = HipEnsemble(T, W, D, like, -box, box, seed=seed)
    eng.upload(np.clip(np.random.RandomState(11).randn(T, W, D), -0.9 * box, 0.9 * box), betas=orc.make_ladder(D, ntemps=T))
    eng.eval_state()
    if mh is not None:
        eng.set_mh_proposal(*mh)
    return eng, mu, invcov


def _reference_draws(rs, T, W):
    """One iteration's draws in the reference's own form (red_blue.py:119-124, stretch.py:93-132, tempering.py:526-541)."""
    N0 = (W + 1) // 2
    d = dict(labels=np.stack([rs.permutation(np.arange(W) % 2) for _ in range(T)]))
    for sp in (0, 1):
        Ns = N0 if sp == 0 else W - N0
        d[f"rint{sp}"] = rs.randint(W - Ns, size=(T, Ns))
        d[f"u_zz{sp}"] = rs.rand(T, Ns)
        d[f"u_acc{sp}"] = rs.rand(T, Ns)
    d["iperm"] = np.stack([rs.permutation(W) for _ in range(T - 1)])
    d["i1perm"] = np.stack([rs.permutation(W) for _ in range(T - 1)])
    d["u_swap"] = rs.rand(T - 1, W)
    return d


def _snapshot(eng, mh=False):
    x, L, P, betas = eng.download()
    c = eng.counters()
    out = dict(x=x, L=L, P=P, betas=betas, accepted=c["accepted"], swaps_total=c["swaps_total"])
    if mh:
        out["accepted_mh"] = eng.mh_counters()["accepted"]
    return out


def _assert_same(a, b, what):
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{what}: {k} differs"


@pytest.mark.parametrize("T,W,D,mh", [(8, 64, 8, None), (16, 128, 32, ("iso", 0.3, 0.5)), (4, 96, 16, ("diag", None, 0.4)),
                                      (2, 128, 64, None)])
def test_call_splitting_is_invisible(T, W, D, mh):
    if mh is not None and mh[1] is None:
        mh = (mh[0], np.full(D, 0.2), mh[2])
    a, *_ = _engine(T, W, D, mh=mh)
    b, *_ = _engine(T, W, D, mh=mh)
    a.step(9)
    for n in (1, 1, 3, 4):
        b.step(n)
    _assert_same(_snapshot(a, mh is not None), _snapshot(b, mh is not None), f"({T},{W},{D}) one call of 9 vs 1+1+3+4")
    a.close()
    b.close()


def test_parity_api_continues_from_a_stepped_state_and_back():
    """step (records, in place) -> teacher-forced half-steps + sweep (by-field arrays, copying) -> step again."""
    T, W, D = 8, 64, 16
    eng, mu, invcov = _engine(T, W, D)
    fn = lambda x: orc.gaussian_log_like(x, mu, invcov)          # noqa: E731
    lo, hi = np.full(D, -50.0), np.full(D, 50.0)
    x, L, P, betas = eng.download()
    st = ru.OracleState(x, L, P, betas)
    it0 = eng.iteration()
    eng.step(3)
    ru.replay(eng, st, it0, 3, fn, lo, hi)
    rs = np.random.RandomState(4)
    for _ in range(2):                                           # the reference's own draws through the parity API
        draws = _reference_draws(rs, T, W)
        for sp in (0, 1):
            eng.stretch_split(sp, draws["labels"], draws[f"rint{sp}"], draws[f"u_zz{sp}"], draws[f"u_acc{sp}"])
        eng.pt_sweep(draws["iperm"], draws["i1perm"], draws["u_swap"], adapt=True)
        ru.oracle_iteration(st, draws, fn, lo, hi)
    x, L, P, betas = eng.download()
    ru.assert_state_equal(st, x, L, P, betas, what="parity API after 3 production iterations")
    it0 = eng.iteration()
    eng.step(4)
    ru.replay(eng, st, it0, 4, fn, lo, hi)
    x, L, P, betas = eng.download()
    ru.assert_state_equal(st, x, L, P, betas, what="production iterations after the parity API")
    eng.close()


_WORKER = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.test_hip_records import _engine, _snapshot
T, W, D, use_mh, like = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
eng, *_ = _engine(T, W, D, mh=("iso", 0.05 if like == "rosen" else 0.3, 0.5) if use_mh else None, like_kind=like)
eng.step(2); eng.step(5)
np.savez(sys.argv[7], **_snapshot(eng, bool(use_mh)))
"""


@pytest.mark.parametrize("T,W,D,use_mh,like", [(16, 256, 32, 0, "dense"), (8, 128, 64, 1, "dense"), (32, 256, 128, 1, "rosen"),
                                              (10, 512, 64, 1, "dense"), (12, 8192, 32, 0, "dense"), (20, 256, 128, 1, "rosen")])
def test_record_mode_equals_the_copying_three_launch_path(T, W, D, use_mh, like, tmp_path):
    outs = []
    for tag, env in (("fused", {}), ("three", {"HENS_NO_FUSED": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        e = dict(os.environ, **env)
        e.pop("HENS_NO_FUSED", None) if tag == "fused" else None
        r = subprocess.run([sys.executable, "-c", _WORKER, ROOT, str(T), str(W), str(D), str(use_mh), like, out],
                           env=e, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(dict(np.load(out)))
    _assert_same(outs[0], outs[1], f"({T},{W},{D}) two-launch record mode vs three copying launches")


def _one_launch(eng, n=3):
    """True when hens_step runs the iterations of this context as single launches (k_iter): no stretch launch of its own."""
    eng.set_profiling(True)
    eng.step(n)
    tm = eng.timing()
    eng.set_profiling(False)
    return tm["n_stretch"] == 0 and tm["n_fused"] == n


@pytest.mark.parametrize("T,W,D,use_mh,like", [(8, 4096, 32, 0, "dense"), (16, 256, 32, 1, "dense"), (4, 1024, 16, 1, "dense"),
                                              (32, 512, 32, 0, "rosen"), (2, 128, 32, 0, "dense"), (10, 2048, 32, 1, "dense"),
                                              (5, 512, 16, 0, "dense")])
def test_one_launch_iteration_equals_the_two_launch_path(T, W, D, use_mh, like, tmp_path):
    """k_iter (hens_iter.h) against k_stretch_fast + k_split1_pt from the same seed: positions, log-probabilities, ladder,
    accept and swap counters bit for bit, across two calls (rows folded back into one half in between) and with the
    Metropolis-Hastings move in the mix (its cascade's counts are adapted stand-alone before the next single launch)."""
    eng, *_ = _engine(T, W, D)
    assert _one_launch(eng), "this shape was expected to step in one launch per iteration"
    eng.close()
    outs = []
    for tag, env in (("one", {}), ("two", {"HENS_NO_ITER": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        e = dict(os.environ, **env)
        if tag == "one":
            e.pop("HENS_NO_ITER", None)
        r = subprocess.run([sys.executable, "-c", _WORKER, ROOT, str(T), str(W), str(D), str(use_mh), like, out],
                           env=e, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(dict(np.load(out)))
    _assert_same(outs[0], outs[1], f"({T},{W},{D}) one launch per iteration vs two")


def test_config2_keeps_the_two_launch_path():
    eng, *_ = _engine(16, 4096, 32)
    assert not _one_launch(eng)
    eng.close()


@pytest.mark.parametrize("knob,T,W,D", [("HENS_NO_COL", 16, 4096, 32), ("HENS_NO_COL", 8, 2048, 64), ("HENS_NO_XCD", 16, 4096, 32),
                                        ("HENS_NO_XCD", 8, 2048, 64), ("HENS_NO_AQL", 16, 4096, 32), ("HENS_NO_AQL", 8, 4096, 32),
                                        ("HENS_NO_AQL", 10, 512, 64), ("HENS_NO_FOLD", 16, 4096, 32), ("HENS_NO_FOLD", 8, 2048, 64),
                                        ("HENS_AQL_FLUSH", 16, 4096, 32), ("HENS_AQL_RELEASE", 16, 4096, 32), ("HENS_AQL_RELEASE", 8, 4096, 32),
                                        ("HENS_AQL_RELEASE", 10, 2048, 32), ("HENS_AQL_RELEASE", 8, 2048, 64)])
def test_every_kept_switch_reaches_the_default_paths_state(knob, T, W, D, tmp_path):
    """The A/B switches the library still reads select another ORDER of the same arithmetic - walker records by slot instead of by
    cascade column, plain instead of XCD-affine workgroup numbering, the HIP stream instead of the context's AQL queue - so the
    chain must be the default path's bit for bit (round 4: every kept switch is in the suite, the others are gone).
    HENS_NO_FOLD on a two-launch shape with the AQL queue on (round 5): the stand-alone adaptation is a HIP-stream kernel between two
    batches of AQL packets - the queue must drain in front of it (fused_iteration), or the ladder is adapted from half-written counts.
    HENS_AQL_FLUSH (any value: "1" = flush register written, never read back) changes how kernel arguments are flushed, nothing else.
    HENS_AQL_RELEASE (round 5): the stepping launches' packets keep their release fence and the records go out as plain stores - the
    default path (no fence: records, row tables and the adaptation's books written through, waves end behind their stores) must
    reach the same state on the two-launch path in column and in slot order and on the one-launch path (see also tools/aql_check.py)."""
    outs = []
    for env in ({}, {knob: "1"}):
        out = str(tmp_path / f"{len(outs)}.npz")
        e = dict(os.environ, **env)
        if not env:
            e.pop(knob, None)
        r = subprocess.run([sys.executable, "-c", _WORKER, ROOT, str(T), str(W), str(D), "0", "dense", out],
                           env=e, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(dict(np.load(out)))
    _assert_same(outs[0], outs[1], f"({T},{W},{D}) default vs {knob}=1")


@pytest.mark.parametrize("T,W,D,like", [(8, 128, 64, "dense"), (32, 256, 128, "rosen")])
def test_cascade_counts_accumulated_or_per_workgroup_same_chain(T, W, D, like, tmp_path):
    """Round 5: the stand-alone cascade of a move mix's MH iterations accumulates its swap counts into a handful of rows that the
    next launch's folded adaptation sums (PtArgs::acc_rows); HENS_PT_NO_ACC=1 keeps a row per workgroup and the k_adapt launch.
    Same counts either way: the chain, the ladder and every counter agree bit for bit."""
    outs = []
    for env in ({}, {"HENS_PT_NO_ACC": "1"}):
        out = str(tmp_path / f"{len(outs)}.npz")
        e = dict(os.environ, **env)
        if not env:
            e.pop("HENS_PT_NO_ACC", None)
        r = subprocess.run([sys.executable, "-c", _WORKER, ROOT, str(T), str(W), str(D), "1", like, out],
                           env=e, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(dict(np.load(out)))
    _assert_same(outs[0], outs[1], f"({T},{W},{D}) MH mix: accumulated counts vs HENS_PT_NO_ACC=1")


# ---- round 6: the fence-free stepping launches' guard inside -m gpu -------------------------------------------
_LONG_WORKER = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.test_hip_records import _engine, _snapshot
T, W, D, use_mh, like, n = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], int(sys.argv[7])
eng, *_ = _engine(T, W, D, like_kind=like)
if use_mh:
    # a call with a Gaussian move in the mix steps on the HIP stream as a whole (hens_step: use_aql): AQL queue -> HIP stream ->
    # AQL queue -> HIP stream -> AQL queue, the walker records handed over in record mode every time
    mh = ("iso", 0.05 if like == "rosen" else 0.3, 0.5)
    eng.step(2); eng.step(n // 4)
    eng.set_mh_proposal(*mh); eng.step(n // 4)
    eng.set_mh_proposal(None, None, 0.0); eng.step(n // 4)
    eng.set_mh_proposal(*mh); eng.step(7)
    eng.set_mh_proposal(None, None, 0.0); eng.step(n - 3 * (n // 4))
else:
    eng.step(2); eng.step(n // 2); eng.step(n - n // 2)
np.savez(sys.argv[8], **_snapshot(eng, bool(use_mh)))
"""
# D = 128 dense; Rosenbrock 4 x 8192 x 128 (one GPU's share of config 5); a short-tile ladder (10 rungs: slot order); a shape that
# steps in ONE launch per iteration (k_iter); calls with and without a Gaussian move in the mix in turn (AQL queue <-> HIP stream)
_LONG_SHAPES = [(8, 1024, 128, 0, "dense"), (4, 8192, 128, 0, "rosen"), (10, 2048, 32, 0, "dense"), (8, 4096, 32, 0, "dense"),
                (16, 1024, 32, 1, "dense"), (4, 2048, 128, 1, "rosen"),
                (8, 16384, 64, 0, "dense")]          # (the config-3 shard: k_stretch2, the persistent first launch, is fence-free too)
_LONG_ITERS = 2000
_long_default = {}


def _long_run(shape, env, tmp_path, tag):
    T, W, D, use_mh, like = shape
    out = str(tmp_path / f"{tag}.npz")
    e = dict(os.environ, **env)
    for k in ("HENS_AQL_RELEASE", "HENS_NO_AQL"):
        if k not in env:
            e.pop(k, None)
    r = subprocess.run([sys.executable, "-c", _LONG_WORKER, ROOT, str(T), str(W), str(D), str(use_mh), like, str(_LONG_ITERS), out],
                       env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return dict(np.load(out))


@pytest.mark.parametrize("knob", ["HENS_AQL_RELEASE", "HENS_NO_AQL"])
@pytest.mark.parametrize("shape", _LONG_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_fence_free_launches_reach_the_fenced_paths_state_over_2000_iterations(knob, shape, tmp_path):
    """The stepping launches of one GPU carry no release fence (hens_aql.h: norel_next): every store a later launch reads is
    written through and every wave ends behind its stores' acknowledgements - an invariant of the kernels' code (its static
    tripwire: tools/store_census.py).  This is its dynamic guard: 2 000 iterations on each shape class the default path serves, against the same
    chain with the fence kept (HENS_AQL_RELEASE=1: plain record stores, release at the end of every packet) and against the HIP
    stream (HENS_NO_AQL=1: the runtime's own fences), final positions, log-probabilities, ladder and counters bit for bit.  One
    stale line read on another XCD anywhere in 2 000 iterations and the chains part."""
    if shape not in _long_default:
        _long_default[shape] = _long_run(shape, {}, tmp_path, "default")
    other = _long_run(shape, {knob: "1"}, tmp_path, knob)
    _assert_same(_long_default[shape], other, f"{shape} default vs {knob}=1 after {_LONG_ITERS} iterations")


@pytest.mark.parametrize("T,W,D,one", [(16, 4096, 32, False), (8, 2048, 64, False), (8, 4096, 32, True)])
def test_dispatch_timestamps_time_the_same_chain_on_the_same_queue(T, W, D, one):
    """hens_set_profiling(ctx, 2) (round 6): per-launch durations from the AQL packets' own dispatch timestamps.  The profiled
    call must (a) stay on the AQL queue (clock == 2), (b) count one first and one second launch per iteration (or one k_iter
    launch), (c) report durations that fit inside the call's begin-to-end span, and (d) leave the chain untouched: the state
    after profiled + plain calls equals the state after plain calls only."""
    a, *_ = _engine(T, W, D)
    b, *_ = _engine(T, W, D)
    a.step(7)
    a.set_profiling(2)
    a.step(20)
    tm = a.timing()
    lt = a.launch_times()
    a.set_profiling(0)
    a.step(5)
    b.step(7); b.step(20); b.step(5)
    _assert_same(_snapshot(a), _snapshot(b), f"({T},{W},{D}) profiled call vs plain call")
    assert tm["clock"] == 2, "the profiled call left the AQL queue"
    assert tm["n_iters"] == 20
    if one:
        assert tm["n_stretch"] == 0 and tm["n_fused"] == 20
    else:
        assert tm["n_stretch"] == 20 and tm["n_fused"] == 20
    n = tm["n_stretch"] + tm["n_fused"]
    assert lt.shape == (n, 2)
    assert np.all(lt[:, 1] > lt[:, 0]) and np.all(lt[1:, 0] >= lt[:-1, 1] - 1e-3), "launches of one queue with barrier bits do not overlap"
    busy = (tm["stretch_ms"] + tm["fused_ms"]) * 1e3
    assert 0 < busy <= tm["total_ms"] * 1e3 * 1.0001
    assert 1.0 < busy / 20 < 200.0, f"implausible per-iteration kernel time {busy / 20:.2f} us"
    a.close(); b.close()


# ---- round 6: the persistent, software-pipelined first launch (hens_tile2.h) ------------------------------------------------------
_TILE2_WORKER = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from tests import parity_utils as pu
from oracle import eryn_oracle as orc
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood, RosenbrockLikelihood
T, W, D, like, n, stop = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
mu, invcov = pu.gaussian_problem(D)
lk = RosenbrockLikelihood(D) if like == "rosen" else GaussianLikelihood(mu, np.diag(invcov).copy() if like == "diag" else invcov)
box = 5.0 if like == "rosen" else 50.0
eng = HipEnsemble(T, W, D, lk, -box, box, seed=5, adaptation_lag=50, adaptation_time=10, stop_adaptation=stop)
eng.upload(np.clip(np.random.RandomState(11).randn(T, W, D), -0.9 * box, 0.9 * box), betas=orc.make_ladder(D, ntemps=T))
eng.eval_state()
eng.step(2); eng.step(n // 2); eng.step(n - n // 2)
x, L, P, betas = eng.download()
c = eng.counters()
eng.set_profiling(2); eng.step(4); tm = eng.timing(); eng.set_profiling(0)
np.savez(sys.argv[8], x=x, L=L, P=P, betas=betas, accepted=c["accepted"], swaps_total=c["swaps_total"])
"""


@pytest.mark.parametrize("T,W,D,like,iters,stop", [(8, 16384, 64, "dense", 60, -1), (4, 512, 64, "dense", 300, -1), (8, 2048, 64, "diag", 300, -1),
                                                    (8, 2048, 64, "rosen", 300, 40), (8, 1168, 64, "dense", 200, 30), (16, 512, 64, "dense", 200, -1),
                                                    (2, 512, 64, "diag", 200, -1)])
def test_persistent_pipelined_first_launch_equals_the_rounds_of_workgroups(T, W, D, like, iters, stop, tmp_path):
    """k_stretch2 (hens_tile2.h; round 6): launches of more than one round of workgroups at D = 64 run the first half-step in
    persistent workgroups that walk two tiles, tile n + 1's phase A and part of its gathers under tile n's likelihood / accept
    phases.  Same arithmetic, same draws: the chain must be k_stretch_fast's (HENS_NO_TILE2=1) bit for bit - at the config-3
    shard (selected by default) and, forced (HENS_TILE2_FORCE=1), on small grids: three likelihoods, a ragged last tile
    (N0 = 584), ladders of 2, 4, 8, 16 rungs, the adaptation moving and stopped half-way (stop_adaptation)."""
    outs = []
    for env in ({"HENS_TILE2_FORCE": "1"}, {"HENS_NO_TILE2": "1"}):
        out = str(tmp_path / f"{len(outs)}.npz")
        e = dict(os.environ, **env)
        for k in ("HENS_TILE2_FORCE", "HENS_NO_TILE2"):
            if k not in env:
                e.pop(k, None)
        r = subprocess.run([sys.executable, "-c", _TILE2_WORKER, ROOT, str(T), str(W), str(D), like, str(iters), str(stop), out],
                           env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(dict(np.load(out)))
    _assert_same(outs[0], outs[1], f"({T},{W},{D},{like}) persistent pipelined first launch vs rounds of workgroups")
    assert outs[0]["accepted"].sum() > 0 and outs[0]["swaps_total"].sum() > 0
