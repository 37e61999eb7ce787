"""AQL dispatch against the HIP stream (HENS_NO_AQL=1) from the same seed: final state and counters must be bit-identical.
  python tools/aql_check.py [T W D n call]      (child mode: ... out.npz)"""
import os, subprocess, sys
import numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)

def chain(T, W, D, n, call):
    import torch  # noqa: F401
    from tools.quick_bench import problem, ladder
    os.environ.pop("HENS_STEP_EVENTS", None)      # (quick_bench sets it on import: an event pair per call, and the HIP stream)
    from eryn_amd.engine import HipEnsemble
    from eryn_amd.likelihood import GaussianLikelihood
    mu, invcov, cov = problem(D)
    eng = HipEnsemble(T, W, D, GaussianLikelihood(mu, invcov), -50.0, 50.0, seed=2024)
    eng.upload(np.random.RandomState(1).randn(T, W, D), betas=ladder(D, T))
    eng.eval_state()
    done = 0
    while done < n:
        k = min(call, n - done); eng.step(k); done += k
        if done % (7 * call) == 0: eng.synchronize()
    x, L, P, b = eng.download()
    c = eng.counters()
    return dict(x=x, L=L, P=P, betas=b, acc=c["accepted"], sw=c["swaps_total"])

if __name__ == "__main__":
    if len(sys.argv) > 6:
        T, W, D, n, call = (int(v) for v in sys.argv[1:6])
        np.savez(sys.argv[6], **chain(T, W, D, n, call))
        sys.exit(0)
    shapes = [(16, 4096, 32, 3000, 20), (16, 4096, 32, 5000, 1777), (4, 512, 32, 2500, 1), (8, 4096, 32, 3000, 20), (10, 2048, 11, 3000, 333),
              (32, 1024, 16, 4000, 7), (8, 16384, 64, 600, 20), (5, 100, 5, 3000, 100)]
    if len(sys.argv) > 5: shapes = [tuple(int(v) for v in sys.argv[1:6])]
    for sh in shapes:
        outs = []
        for tag, env in (("aql", {}), ("hip", {"HENS_NO_AQL": "1"})):
            out = f"/tmp/aqlchk_{tag}.npz"
            r = subprocess.run([sys.executable, __file__] + [str(v) for v in sh] + [out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
            if r.returncode: print(tag, "FAILED", r.stderr[-1500:]); sys.exit(1)
            if "AQL" in r.stderr: print(r.stderr[-500:])
            outs.append(dict(np.load(out)))
        bad = [k for k in outs[0] if not np.array_equal(outs[0][k], outs[1][k])]
        print(sh, "AQL == HIP stream bit for bit" if not bad else f"DIFFERS in {bad}", flush=True)
        if bad: sys.exit(2)
