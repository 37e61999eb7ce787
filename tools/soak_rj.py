"""Repeat a config-4 RJ chain (all three birth / death schedules) and compare the repeats: the production RJ path must be a
pure function of the seed.   python tools/soak_rj.py [iters] [reps]"""
import sys, os, hashlib, subprocess
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
if len(sys.argv) > 3:
    import numpy as np
    from eryn_amd.moves.tempering import make_ladder
    from eryn_amd.rj import RJEngine, TemplateBranch
    iters, sched = int(sys.argv[1]), int(sys.argv[3])
    T, W, N, NL = 8, 2048, 500, 10
    t = np.linspace(-1, 1, N); rs = np.random.RandomState(42)
    gi = np.array([[3.3, -0.2, 0.1], [2.6, -0.1, 0.1], [3.4, 0.0, 0.1], [2.9, 0.3, 0.1]]); si = np.array([[1.3, 10.1, 1.0], [0.8, 4.6, 1.2]])
    y = sum(a * np.exp(-((t - b) ** 2) / (2 * c ** 2)) for a, b, c in gi) + sum(a * np.sin(2 * np.pi * b * t + c) for a, b, c in si) + 2.0 * rs.randn(N)
    brs = [TemplateBranch("gauss", "pulse", [(2.5, 3.5), (-1.0, 1.0), (0.01, 0.21)], NL, 0),
           TemplateBranch("sine", "sine", [(0.5, 1.5), (1.0, 20.0), (0.0, 2 * np.pi)], NL, 0)]
    eng = RJEngine(T, W, brs, t, y, 2.0, seed=2024)
    x = {"gauss": np.zeros((T, W, NL, 3)), "sine": np.zeros((T, W, NL, 3))}
    inds = {k: np.zeros((T, W, NL), dtype=bool) for k in x}
    for n in range(4):
        x["gauss"][:, :, n] = gi[n] + 1e-2 * rs.randn(T, W, 3) * [1, 1, 0.1]; inds["gauss"][:, :, n] = True
    for n in range(2):
        x["sine"][:, :, n] = si[n] + 1e-2 * rs.randn(T, W, 3); inds["sine"][:, :, n] = True
    eng.upload(x, inds, betas=make_ladder(18, ntemps=T)); eng.eval_state()
    eng.set_mh_scale(np.full((2, 3), 1e-2) * [[1, 1, 0.1], [1, 1, 1]])
    eng.set_schedule(("separate_branches", "iterate_branches", "together")[sched])
    done = 0
    while done < iters:
        k = min(333, iters - done); eng.step(k); done += k
    x1, i1, L, P, b = eng.download(); c = eng.counters()
    h = hashlib.sha1()
    for a in (x1["gauss"], x1["sine"], i1["gauss"], i1["sine"], L, P, b): h.update(np.ascontiguousarray(a).tobytes())
    for k in sorted(c):
        if hasattr(c[k], "tobytes"): h.update(np.ascontiguousarray(c[k]).tobytes())
    print("HASH", h.hexdigest()[:12], float(sum(v.sum() for v in i1.values())) / (T * W))
else:
    iters = sys.argv[1] if len(sys.argv) > 1 else "3000"; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    for sched in (0, 1, 2):
        hs = []
        for r in range(reps):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), iters, "x", str(sched)], capture_output=True, text=True)
            hs.append([l for l in p.stdout.splitlines() if l.startswith("HASH")][-1] if "HASH" in p.stdout else "ERR " + p.stderr[-200:])
        print(f"schedule {sched}: {iters} iterations x {reps} repeats: {'all the same' if len(set(hs)) == 1 else 'DIFFER'}  {hs[0]}", flush=True)
