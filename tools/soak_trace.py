"""Per-chunk field hashes of one chain, for locating where two runs of the same chain part ways.
  python tools/soak_trace.py T W D n chunk mh out.npz        (env knobs select the path)"""
import sys, os, hashlib, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from tools.quick_bench import problem, ladder
from eryn_amd.engine import HipEnsemble
from eryn_amd.likelihood import GaussianLikelihood
T, W, D, n, chunk, mh = (int(v) for v in sys.argv[1:7])
mu, invcov, cov = problem(D)
eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=99)
eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T)); eng.eval_state()
if mh: eng.set_mh_proposal("iso", 0.3, 0.3)
h = lambda a: int(hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12], 16)
rows = []
snap = {}
done = 0
while done < n:
    eng.step(chunk); done += chunk
    x, L, P, b = eng.download(); c = eng.counters()
    rows.append([h(x), h(L), h(b), h(c["accepted"]), h(c["swaps_total"])])
    snap = dict(x=x, L=L, b=b, acc=c["accepted"], sw=c["swaps_total"])
    if os.environ.get("SOAK_KEEP") and done in (int(v) for v in os.environ["SOAK_KEEP"].split(",")):
        np.savez(sys.argv[7] + f".at{done}.npz", **snap)
np.savez(sys.argv[7], rows=np.array(rows, dtype=np.uint64), **snap)
