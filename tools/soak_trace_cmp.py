import sys, os, subprocess, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", ".")
T, W, D, n, chunk, mh, reps = sys.argv[1:8]
env = dict(x.split(":") for x in sys.argv[8].split(",")) if len(sys.argv) > 8 and sys.argv[8] else {}
outs = []
for r in range(int(reps)):
    out = f"/tmp/trace_{r}.npz"
    p = subprocess.run([sys.executable, os.path.join(root, "tools/soak_trace.py"), T, W, D, n, chunk, mh, out], env=dict(os.environ, **env), capture_output=True, text=True)
    if p.returncode: print(p.stderr[-500:])
    outs.append(np.load(out))
ref = outs[0]["rows"]
names = ["x", "L", "betas", "accepted", "swaps_total"]
for r, o in enumerate(outs[1:], 1):
    d = o["rows"] != ref
    if not d.any(): print(f"run {r}: identical to run 0 in all {len(ref)} chunks"); continue
    first = int(np.argmax(d.any(axis=1)))
    print(f"run {r}: first difference in chunk {first} (iteration <= {(first + 1) * int(chunk)}): fields {[n_ for n_, f in zip(names, d[first]) if f]}; next chunk: {[n_ for n_, f in zip(names, d[min(first + 1, len(d) - 1)]) if f]}")
