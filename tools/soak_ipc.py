"""Long run of the ladder pipeline with one PROCESS per rank (mailboxes mapped over HIP IPC, as on a multi-GPU node; here the
processes share one GPU) against one context.   python tools/soak_ipc.py [world T W D iters delay]"""
import os, subprocess, sys, tempfile
import numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", ".")
WORKER = os.path.join(root, "tests", "pipeline_worker.py")
KEYS = ("x", "L", "P", "betas", "accepted", "swaps_total", "swaps_last")
world, T, W, D, iters, delay = (int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (2, 8, 512, 32, 4000, 0)))
env = dict(os.environ, PIPE_TEST_DELAY=str(delay), PIPE_TEST_MODEL="gauss", GPU_MAX_HW_QUEUES="16", HENS_PIPE_TIMEOUT_S="30",
           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000))
tmp = tempfile.mkdtemp()
r = subprocess.run([sys.executable, WORKER, "single", str(T), str(W), str(D), str(iters), os.path.join(tmp, "single.npz")], env=env, capture_output=True, text=True, timeout=600)
assert r.returncode == 0, r.stderr[-400:]
procs = [subprocess.Popen([sys.executable, WORKER, "ipc", str(k), str(world), str(T), str(W), str(D), str(iters), tmp], env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for k in range(world)]
outs = [p.communicate(timeout=900)[0] for p in procs]
assert all(p.returncode == 0 for p in procs), "\n".join(o[-400:] for o in outs)
ref = np.load(os.path.join(tmp, "single.npz"))
snaps = [np.load(os.path.join(tmp, f"rank{k}.npz")) for k in range(world)]
got = {k: np.concatenate([s[k] for s in snaps], axis=0) for k in ("x", "L", "P", "accepted")}
for k in ("betas", "swaps_total", "swaps_last"): got[k] = snaps[0][k]
bad = [k for k in KEYS if not np.array_equal(ref[k], got[k])]
print(f"{world} processes, {T}x{W}x{D}, {iters} iterations, delay {delay}: IPC pipeline == one context: {not bad} {bad}")
