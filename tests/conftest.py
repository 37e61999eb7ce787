import os
import sys

import pytest

try:                       # torch first: a process that uses both must load torch's HIP runtime before libhipensemble
    import torch  # noqa: F401  (eryn_amd/_lib.py), whatever subset of the tests is selected
except ImportError:        # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_device():
    try:
        from eryn_amd import _lib
        return _lib.load().hens_device_count() > 0
    except Exception:                          # library missing / broken: do NOT skip - let the gpu tests fail loudly
        return True


def pytest_collection_modifyitems(config, items):
    """Plain `pytest tests` on a box without an MI355X skips the gpu-marked tests instead of failing them
    (the product itself still fails loudly without a device: eryn_amd has no CPU fallback)."""
    if any("gpu" in item.keywords for item in items) and not _have_device():
        skip = pytest.mark.skip(reason="no HIP device visible (gpu tests run on the MI355X box)")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """what the log-likelihood comparisons of this session actually observed (tests/tolerance_log.py)."""
    from tests import tolerance_log as tol
    rep = tol.report()
    if not rep:
        return
    out = tol.write(os.path.join(ROOT, "gpurun_out", "tolerance_report.json"))
    tr = terminalreporter
    tr.write_sep("-", f"log-likelihood vs oracle: worst relative difference {out['worst']:.2e} (bar {tol.RTOL_L:.0e}; per test in gpurun_out/tolerance_report.json)")
    worst = sorted(rep.items(), key=lambda kv: -kv[1]["max_rel_L"])[:8]
    for name, v in worst:
        tr.write_line(f"  {v['max_rel_L']:.2e}  (bar {v['rtol']:.0e}, {v['values']} values)  {name}")
