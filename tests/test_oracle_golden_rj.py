"""Pin oracle/eryn_oracle_rj.py (the reversible-jump leaf-packing restatement) to the reference.

tests/golden/make_golden_rj.py ran the REAL reference on three scenarios and recorded the state after every in-model
move and every RJ move.  The oracle gets only the configuration, the initial state and the two seeds; it must land on
every recorded array BIT FOR BIT (coordinates of every leaf slot - dead ones included -, inds, log-like, log-prior,
accept masks, betas, swap counts), which pins the arithmetic and the order of every draw of both streams."""
import os

import numpy as np
import pytest

from oracle import eryn_oracle_rj as orj

NAMES = ["rj1_two_branches", "rj2_min_leaves", "rj3_ten_leaves"]
# rj_moves="iterate_branches": one RJ move walks through every branch; "together": one proposal changes every branch at once
NAMES_ALL = NAMES + ["rj4_iterate_branches", "rj5_together"]
# round 5: the red / blue StretchMove as the in-model move over every branch and leaf slot (no reversible jump / beside it)
NAMES_STRETCH = ["rjs1_stretch_fixed_leaves", "rjs2_stretch_with_rj"]
# round 6: the likelihood is a plain Python function of the packed active leaves (oracle/eryn_oracle_rj.py: lorentz_chirp_log_like),
# not the template model - separate_branches, "together" with a leaf floor, the stretch move as the in-model move, "iterate_branches"
NAMES_CALLABLE = ["rjh1_callable", "rjh2_callable_together", "rjh3_callable_stretch", "rjh4_callable_iterate"]
# ... over branches of OTHER leaf widths than three (ndims = {ramp: 2, burst: 4}; one branch of one-parameter leaves, where the reference
# hands the function the leaves themselves instead of a list over the branches)
NAMES_WIDTHS = ["rjn1_widths_2_4", "rjn2_widths_together", "rjn3_widths_stretch", "rjn4_width_1_iterate"]
LIKES = {"lorentz_chirp": "lorentz_chirp_log_like", "ramp_burst": "ramp_burst_log_like", "offset": "offset_log_like"}


def load_rj(golden_dir, name):
    with np.load(os.path.join(golden_dir, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


def make_rj_oracle(fx, record=False):
    if "branch_names" in fx:                                   # (a model of general leaf widths: the boxes say how wide)
        branches = [orj.Branch(str(k), orj.KIND_PULSE, fx[f"{k}_box"], int(fx["nl_max"][i]), int(fx["nl_min"][i]),
                               np.diag(np.ones(len(fx[f"{k}_box"]))) * float(fx["cov_factor"])) for i, k in enumerate(fx["branch_names"])]
    else:
        cov = np.diag(np.ones(3)) * float(fx["cov_factor"])
        branches = [orj.Branch("gauss", orj.KIND_PULSE, fx["gauss_box"], int(fx["nl_max"][0]), int(fx["nl_min"][0]), cov),
                    orj.Branch("sine", orj.KIND_SINE, fx["sine_box"], int(fx["nl_max"][1]), int(fx["nl_min"][1]), cov)]
    R = np.random.RandomState(int(fx["seed_construct"]))      # R := snapshot of the global stream at construction
    G = np.random.RandomState(int(fx["seed_run"]))
    x0 = {b.name: fx[f"x0_{b.name}"] for b in branches}
    inds0 = {b.name: fx[f"inds0_{b.name}"] for b in branches}
    return orj.OracleRJSampler(branches, x0, inds0, fx["t"], fx["y"], float(fx["sigma"]), R, G, fx["betas0"],
                               record=record, schedule=str(fx["rj_moves"]) if "rj_moves" in fx else "separate_branches",
                               in_model=str(fx["in_model"]) if "in_model" in fx else "gaussian",
                               like_fn=getattr(orj, LIKES[str(fx["model"])]) if "model" in fx else None)


@pytest.mark.parametrize("name", NAMES_ALL + NAMES_STRETCH + NAMES_CALLABLE + NAMES_WIDTHS)
def test_rj_oracle_reproduces_the_reference(golden_dir, name):
    fx = load_rj(golden_dir, name)
    o = make_rj_oracle(fx)
    assert np.array_equal(o.st.P, fx["P0"]) and np.array_equal(o.st.L, fx["L0"])
    for it in range(int(fx["nsteps"])):
        acc = o.mh_move() if o.in_model == "gaussian" else o.stretch_move()
        pre = f"it{it}_mh_"
        assert np.array_equal(acc, fx[pre + "accepted"]), f"{pre}accepted"
        for b in o.branches:
            assert np.array_equal(o.st.x[b.name], fx[pre + f"x_{b.name}"]), f"{pre}x_{b.name}"
            assert np.array_equal(o.st.inds[b.name], fx[pre + f"inds_{b.name}"])
        assert np.array_equal(o.st.L, fx[pre + "L"]) and np.array_equal(o.st.P, fx[pre + "P"]), pre
        assert np.array_equal(o.st.betas, fx[pre + "betas"]) and np.array_equal(o.swaps_accepted, fx[pre + "swaps"])
        if o.schedule == "none":
            continue
        bi, racc = o.rj_move()
        pre = f"it{it}_rj_"
        assert bi == int(fx[pre + "branch"])
        assert np.array_equal(racc, fx[pre + "accepted"]), f"{pre}accepted"
        for b in o.branches:
            assert np.array_equal(o.st.x[b.name], fx[pre + f"x_{b.name}"]), f"{pre}x_{b.name}"
            assert np.array_equal(o.st.inds[b.name], fx[pre + f"inds_{b.name}"]), f"{pre}inds_{b.name}"
        assert np.array_equal(o.st.L, fx[pre + "L"]) and np.array_equal(o.st.P, fx[pre + "P"]), pre
        assert np.array_equal(o.st.betas, fx[pre + "betas"]) and np.array_equal(o.swaps_accepted, fx[pre + "swaps"])
    assert np.array_equal(o.mh_accepted, fx["mh_accepted_total"])
    if o.schedule == "none":
        assert fx["mh_accepted_total"].sum() > 0 and not all(fx[f"inds0_{b.name}"].all() for b in o.branches), "some slots stay dead"
        return
    nm = fx["rj_accepted_total"].shape[0]                     # one move object per branch, or ONE for "iterate_branches"
    assert np.array_equal(np.stack(o.rj_accepted)[:nm], fx["rj_accepted_total"])
    assert np.array_equal(np.array(o.rj_num_proposals)[:nm], fx["rj_num_proposals"])
    # the scenarios do what they are there for
    assert fx["rj_accepted_total"].sum() > 0 and fx["mh_accepted_total"].sum() > 0
