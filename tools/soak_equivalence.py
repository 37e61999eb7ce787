import sys, os, numpy as np, subprocess
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
W = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from tools.quick_bench import problem, ladder
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood
T, W, D, n, mh = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
mu, invcov, cov = problem(D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=99)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T)); eng.eval_state()
if mh: eng.set_mh_proposal("iso", 0.3, 0.3)
done = 0
while done < n:
    k = min(7777, n - done); eng.step(k); done += k
x, L, P, b = eng.download(); c = eng.counters()
np.savez(sys.argv[7], x=x, L=L, P=P, b=b, acc=c["accepted"], sw=c["swaps_total"])
print("finite", np.isfinite(x).all(), np.isfinite(L).all(), "acc", c["accepted"].mean() / max(c["num_proposals"], 1))
'''
root = os.environ.get("GRAFT_REPO_ROOT", ".")
if len(sys.argv) > 1 and sys.argv[1] == "iter":
    # the one-launch iteration (k_iter: versioned rows, replayed complements, three count buffers, short tiles, padded rows)
    # against the two launches from the same seed
    for (T, Wk, D, n, mh) in ((8, 4096, 32, 200000, 0), (10, 2048, 11, 100000, 1), (16, 4096, 16, 100000, 1)):
        outs = []
        for tag, env in (("one", {}), ("two", {"HENS_NO_ITER": "1"})):
            out = f"/tmp/soak_{tag}.npz"
            r = subprocess.run([sys.executable, "-c", W, root, str(T), str(Wk), str(D), str(n), str(mh), out], env=dict(os.environ, **env), capture_output=True, text=True)
            print(tag, r.stdout.strip()[-200:], r.stderr.strip()[-300:])
            outs.append(dict(np.load(out)))
        same = all(np.array_equal(outs[0][k], outs[1][k]) for k in outs[0])
        print(f"({T},{Wk},{D}) {n} iterations mh={mh}: one launch per iteration == two launches:", same)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "xcd":
    # first launch with workgroups renumbered so that an XCD works on whole rungs (its L2 keeps rows from launch to launch)
    # against the plain numbering (HENS_NO_XCD=1): a stale line in one XCD's L2 would show up as a different chain
    for (T, Wk, D, n, mh) in ((16, 4096, 32, 200000, 0), (8, 16384, 64, 20000, 0), (16, 4096, 32, 60000, 1), (4, 2048, 16, 100000, 0)):
        outs = []
        for tag, env in (("xcd", {}), ("plain", {"HENS_NO_XCD": "1"})):
            out = f"/tmp/soak_{tag}.npz"
            r = subprocess.run([sys.executable, "-c", W, root, str(T), str(Wk), str(D), str(n), str(mh), out], env=dict(os.environ, **env), capture_output=True, text=True)
            print(tag, r.stdout.strip()[-200:], r.stderr.strip()[-300:])
            outs.append(dict(np.load(out)))
        same = all(np.array_equal(outs[0][k], outs[1][k]) for k in outs[0])
        print(f"({T},{Wk},{D}) {n} iterations mh={mh}: XCD-affine first launch == plain numbering:", same)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "col":
    # column-ordered records (round 3: coalesced record loads, next buffers written in the NEXT iteration's column order by
    # scattered stores) against slot-ordered records (HENS_NO_COL=1) from the same seed; chunks of 7777 iterations also cross
    # the 1024-iteration key windows at every offset
    for (T, Wk, D, n) in ((16, 4096, 32, 200000), (64, 2048, 64, 30000), (32, 8192, 32, 40000), (2, 16384, 8, 60000)):
        outs = []
        for tag, env in (("col", {}), ("slot", {"HENS_NO_COL": "1"})):
            out = f"/tmp/soak_{tag}.npz"
            r = subprocess.run([sys.executable, "-c", W, root, str(T), str(Wk), str(D), str(n), "0", out], env=dict(os.environ, **env), capture_output=True, text=True)
            print(tag, r.stdout.strip()[-200:], r.stderr.strip()[-300:])
            outs.append(dict(np.load(out)))
        same = all(np.array_equal(outs[0][k], outs[1][k]) for k in outs[0])
        print(f"({T},{Wk},{D}) {n} iterations: column-ordered records == slot-ordered records:", same)
    sys.exit(0)
for (T, Wk, D, n, mh) in ((16, 4096, 32, 200000, 0), (8, 2048, 64, 60000, 1), (32, 1024, 16, 100000, 1)):
    outs = []
    for tag, env in (("fused", {}), ("three", {"HENS_NO_FUSED": "1"})):
        out = f"/tmp/soak_{tag}.npz"
        r = subprocess.run([sys.executable, "-c", W, root, str(T), str(Wk), str(D), str(n), str(mh), out], env=dict(os.environ, **env), capture_output=True, text=True)
        print(tag, r.stdout.strip()[-200:], r.stderr.strip()[-300:])
        outs.append(dict(np.load(out)))
    same = all(np.array_equal(outs[0][k], outs[1][k]) for k in outs[0])
    print(f"({T},{Wk},{D}) {n} iterations mh={mh}: record mode == three copying launches:", same)
