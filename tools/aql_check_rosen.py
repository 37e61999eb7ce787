"""The stepping launches without a release fence (default) against HENS_AQL_RELEASE=1 and HENS_NO_AQL=1 on Rosenbrock shapes
(tools/aql_check.py covers the dense Gaussian): final state and counters bit for bit.  Uses the worker of tests/test_hip_records.py."""
import os, subprocess, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from tests import test_hip_records as t
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
for (T, W, D, like) in [(4, 2048, 128, "rosen"), (8, 4096, 64, "rosen"), (16, 1024, 32, "rosen"), (8, 2048, 16, "rosen")]:
    outs = []
    for env in ({}, {"HENS_AQL_RELEASE": "1"}, {"HENS_NO_AQL": "1"}):
        out = f"/tmp/rc_{len(outs)}.npz"
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, "-c", t._WORKER, ROOT, str(T), str(W), str(D), "0", like, out], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(dict(np.load(out)))
    t._assert_same(outs[0], outs[1], "default vs HENS_AQL_RELEASE")
    t._assert_same(outs[0], outs[2], "default vs HENS_NO_AQL")
    print((T, W, D, like), "no-release path == fence kept == HIP stream, bit for bit")
