"""Build libhipensemble.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.environ.get("HENS_LIB") or os.path.join(LIB_DIR, "libhipensemble.so")
OBJ_DIR = os.environ.get("HENS_OBJ_DIR") or os.path.join(os.path.dirname(HERE), "build", "hens_obj")      # (A/B libraries: objects of their own)
# hens.hip: the C ABI, the host logic and the small kernels; hens_k_<likelihood>.hip (x 2 parts): the stepping kernels' instantiations
# (csrc/hens_ktable.h) - one translation unit per (likelihood, part), compiled side by side, then linked.
UNITS = [("hens", "hens.hip", [])] + [(f"hens_k_{k}_{part}", f"hens_k_{k}.hip", [f"-DHENS_KT_PART={part}"])
                                      for k in ("dense", "diag", "rosen") for part in (0, 1)]
SOURCES = sorted({os.path.join(SRC_DIR, u[1]) for u in UNITS})
DEPS = SOURCES + [os.path.join(SRC_DIR, h) for h in ("hens_kernels.h", "hens_rj.h", "hens_iter.h", "hens_tile2.h", "hens_aql.h", "hens_ktable.h", "hens_ktable.inc")] + [
    os.path.join(os.path.dirname(HERE), "include", "hipensemble.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-cuda-compat"] + os.environ.get("HENS_BUILD_DEFS", "").split()   # (inline __global__: see hens_kernels.h)


def link_flags(hipcc):
    """-lhsa-runtime64 (direct AQL dispatch of the stepping launches: csrc/hens_aql.h) from the ROCm tree hipcc itself lives in."""
    root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
    dirs = [d for d in (os.path.join(root, "lib"), "/opt/rocm/lib") if os.path.isdir(d)]
    return ["-L" + d for d in dict.fromkeys(dirs)] + ["-lhsa-runtime64"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    """Compile the HIP extension if missing or older than its sources.  Returns the .so path."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = hipcc_path()
    include = ["-I" + os.path.join(os.path.dirname(HERE), "include")]
    jobs = []
    for name, src, defs in UNITS:
        obj = os.path.join(OBJ_DIR, name + ".o")
        cmd = [hipcc] + FLAGS + defs + include + ["-c", os.path.join(SRC_DIR, src), "-o", obj]
        # an object is kept if it is newer than everything its unit includes (the kernel units see neither hens.hip nor the RJ / AQL
        # headers) and was made by the same command
        deps = [os.path.join(SRC_DIR, src), os.path.join(os.path.dirname(HERE), "include", "hipensemble.h"), os.path.abspath(__file__)] + [
            os.path.join(SRC_DIR, h) for h in (("hens_kernels.h", "hens_iter.h", "hens_tile2.h", "hens_ktable.h") +
                                               (("hens_rj.h", "hens_aql.h") if name == "hens" else ("hens_ktable.inc",)))]
        stamp = obj + ".cmd"
        fresh = (os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == " ".join(cmd) and
                 all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps))
        if fresh and not os.environ.get("HENS_BUILD_ALL"):
            jobs.append((obj, cmd, None, stamp))
            continue
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((obj, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), stamp))
    failed = []
    for obj, cmd, proc, stamp in jobs:     # (7 compilers side by side: about 1.5 GB each at their peak)
        if proc is None:
            continue
        out, _ = proc.communicate()
        if proc.returncode != 0:
            failed.append(" ".join(cmd) + "\n" + out)
            if os.path.exists(stamp):
                os.remove(stamp)
        else:
            with open(stamp, "w") as f:
                f.write(" ".join(cmd))
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [j[0] for j in jobs] + ["-o", LIB_PATH] + link_flags(hipcc)
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
