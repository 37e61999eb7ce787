"""Repeat one soak shape several times per variant and compare everything with the first run (hunting a rare divergence).
  python tools/soak_one.py T W D n mh reps [variant=ENV=VAL,...]"""
import sys, os, numpy as np, subprocess
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
src = open(os.path.join(root, "tools/soak_equivalence.py")).read()
W = src.split("W = r'''")[1].split("'''")[0]
T, Wk, D, n, mh, reps = (int(v) for v in sys.argv[1:7])
def run(tag, env):
    out = f"/tmp/soak_{tag}.npz"
    r = subprocess.run([sys.executable, "-c", W, root, str(T), str(Wk), str(D), str(n), str(mh), out], env=dict(os.environ, **env), capture_output=True, text=True)
    if r.returncode: print(tag, r.stderr[-400:])
    return dict(np.load(out))
variants = {"fused": {}, "three": {"HENS_NO_FUSED": "1"}}
for a in sys.argv[7:]:
    name, kv = a.split("=", 1)
    variants[name] = dict(x.split(":") for x in kv.split(","))
ref = None
for rep in range(reps):
    for k, e in variants.items():
        v = run(k, e)
        if ref is None: ref = v
        bad = [f for f in v if not np.array_equal(v[f], ref[f])]
        print(rep, k, "same" if not bad else f"DIFFERS in {bad}", flush=True)
