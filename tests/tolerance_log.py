"""Observed floating-point differences between the HIP path and the oracle, per test (log-likelihood rel 1e-13).

Every comparison of log-likelihoods goes through ``check_logl``: it records the largest relative difference it saw under the
running test's name and asserts it against the bar.  ``conftest.pytest_terminal_summary`` prints the table and writes it to
``gpurun_out/tolerance_report.json`` (copied to ``profiles/`` per round)."""
import json
import os

import numpy as np

RTOL_L = 1e-13          # SURVEY 8c's bar for log-likelihoods (summation order of the quadratic form); exceptions are argued where made
_seen = {}


def _test_name():
    return os.environ.get("PYTEST_CURRENT_TEST", "outside pytest").split(" ")[0]


def max_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = np.abs(a - b) / np.abs(b)
    r = np.where(a == b, 0.0, r)                       # (equal infinities / fill values / zeros)
    r = r[np.isfinite(r)]
    return float(r.max()) if r.size else 0.0


def check_logl(L, Lref, rtol=RTOL_L, what=""):
    rel = max_rel(L, Lref)
    e = _seen.setdefault(_test_name(), {"max_rel_L": 0.0, "rtol": rtol, "comparisons": 0, "values": 0})
    e["max_rel_L"] = max(e["max_rel_L"], rel)
    e["rtol"] = max(e["rtol"], rtol)
    e["comparisons"] += 1
    e["values"] += int(np.size(Lref))
    same_special = np.array_equal(np.isfinite(L), np.isfinite(Lref))
    assert same_special and rel <= rtol, f"{what}: log-likelihood differs from the oracle by {rel:.3e} relative (bar {rtol:.0e})"
    return rel


def report():
    return dict(sorted(_seen.items()))


def write(path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rep = report()
    out = {"bar": RTOL_L, "worst": max([v["max_rel_L"] for v in rep.values()] + [0.0]), "tests": rep}
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    return out
