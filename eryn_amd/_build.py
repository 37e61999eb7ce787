"""Build libhipensemble.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.environ.get("HENS_LIB") or os.path.join(LIB_DIR, "libhipensemble.so")
SOURCES = [os.path.join(SRC_DIR, "hens.hip")]
DEPS = SOURCES + [os.path.join(SRC_DIR, "hens_kernels.h"), os.path.join(SRC_DIR, "hens_rj.h"), os.path.join(SRC_DIR, "hens_iter.h"), os.path.join(SRC_DIR, "hens_aql.h"),
                  os.path.join(os.path.dirname(HERE), "include", "hipensemble.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]
LINK = ["-L/opt/rocm/lib", "-lhsa-runtime64"]      # (direct AQL dispatch of the stepping launches: csrc/hens_aql.h)


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
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc_path()] + FLAGS + SOURCES + ["-o", LIB_PATH] + LINK
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
