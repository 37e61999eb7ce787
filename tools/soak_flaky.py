"""How often does a chain differ from the majority of its repeats?  python tools/soak_flaky.py T W D n mh reps name=ENV:VAL,... ..."""
import sys, os, numpy as np, subprocess, collections, hashlib
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
src = open(os.path.join(root, "tools/soak_equivalence.py")).read()
W = src.split("W = r'''")[1].split("'''")[0]
T, Wk, D, n, mh, reps = (int(v) for v in sys.argv[1:7])
variants = {}
for a in sys.argv[7:]:
    name, kv = a.split("=", 1)
    variants[name] = dict(x.split(":") for x in kv.split(",")) if kv else {}
def run(tag, env):
    out = f"/tmp/soak_{tag}.npz"
    r = subprocess.run([sys.executable, "-c", W, root, str(T), str(Wk), str(D), str(n), str(mh), out], env=dict(os.environ, **env), capture_output=True, text=True)
    if r.returncode: print(tag, r.stderr[-300:]); return "error"
    v = np.load(out)
    return hashlib.sha1(b"".join(np.ascontiguousarray(v[f]).tobytes() for f in sorted(v.files))).hexdigest()[:10]
res = collections.defaultdict(list)
for rep in range(reps):
    for k, e in variants.items():
        res[k].append(run(k, e))
allh = collections.Counter(h for v in res.values() for h in v)
major = allh.most_common(1)[0][0]
for k, v in res.items():
    print(f"{k:28s} {sum(h != major for h in v):2d} of {len(v)} differ from the majority   {collections.Counter(v).most_common(3)}", flush=True)
