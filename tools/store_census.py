"""Static census of the global stores in the stepping kernels that run WITHOUT a release fence on their AQL packet.

The fence-free launches of one GPU (k_stretch_fast / k_stretch2 / k_split1_pt / k_iter, non-PIPE instantiations; hens_kernels.h:
wt_store, launch_end_wait) are correct only while every store a LATER launch reads is write-through (`sc1`, or `sc0 sc1`) - a plain
store may sit dirty in one XCD's L2 when the launch ends.  Those kernels also carry the FENCED form of the same stores behind a
run-time branch (HENS_AQL_RELEASE=1, the HIP-stream path) and a few stores nothing on the device reads back (the accept mask for the
host), so "no plain store at all" cannot be asserted from the ISA; what can is that their NUMBER does not change unnoticed: this tool
counts, per kernel, global / flat / buffer stores, the plain ones among them (no sc0 / sc1 / nt bit) and the atomics, and
tests/test_host_logic.py compares the census of the built library with tests/golden/store_census.json.  A kernel change that moves a
count fails that test: review the new store (does a later launch read it?  then wt_store / store_row16), then

    python tools/store_census.py --write        # regenerates tests/golden/store_census.json
    python tools/store_census.py [substring]    # prints the census (plain stores listed with --list)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("HENS_LIB") or os.path.join(ROOT, "eryn_amd", "lib", "libhipensemble.so")
LLVM = "/opt/rocm/lib/llvm/bin"
GOLDEN = os.path.join(ROOT, "tests", "golden", "store_census.json")
STORE = re.compile(r"\t(global|flat|buffer)_store")
ATOMIC = re.compile(r"\t(global|flat|buffer)_atomic")


def fence_free(name):
    """The instantiations hens.hip launches with norel = 1 (norel_ok: one GPU, AQL queue): template arguments by position."""
    m = re.match(r"void hens::(k_\w+)<([^>]*)>", name)
    if not m:
        return False
    k, a = m.group(1), [s.strip() for s in m.group(2).split(",")]
    if k == "k_stretch_fast":            # <DT, LIKE, MODE, NW, PIPE, PER>: the stretch half-step (MODE 0) off a pipeline rank
        return a[2] == "0" and a[4] == "false"
    if k == "k_stretch2":                # <DT, LIKE, PIPE>
        return a[2] == "false"
    if k == "k_split1_pt":               # <DT, LIKE, NW, PER, SHORT, PIPE, COL>
        return a[5] == "false"
    return k == "k_iter"


def census(lib=LIB, want_lines=False):
    out = {}
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "lib.so")
        os.symlink(os.path.abspath(lib), so)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], cwd=d, capture_output=True, check=True)
        for co in sorted(f for f in os.listdir(d) if "gfx950" in f):
            txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(d, co)], capture_output=True, text=True, check=True).stdout
            chunks = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", txt)
            syms = [re.match(r"[0-9a-f]+ <([^>]+)>:", c) for c in chunks]
            names = subprocess.run(["c++filt"], input="\n".join(s.group(1) if s else "-" for s in syms), capture_output=True, text=True).stdout.splitlines()
            for c, s, n in zip(chunks, syms, names):
                if not s or not fence_free(n):
                    continue
                lines = [ln.split("//")[0].strip() for ln in c.split("\n")]
                st = [ln for ln in lines if STORE.search("\t" + ln)]
                plain = [ln for ln in st if not re.search(r"\b(sc0|sc1|nt)\b", ln)]
                e = {"stores": len(st), "plain": len(plain), "atomics": sum(1 for ln in lines if ATOMIC.search("\t" + ln))}
                if want_lines:
                    e["plain_lines"] = plain
                out[n.replace("void hens::", "").split("(")[0]] = e
    return dict(sorted(out.items()))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    c = census(want_lines="--list" in sys.argv)
    if "--write" in sys.argv:
        with open(GOLDEN, "w") as f:
            json.dump({"toolchain": "ROCm 7.2.0 hipcc, -O3 --offload-arch=gfx950 (eryn_amd/_build.py)", "kernels": c}, f, indent=1)
        print(f"{len(c)} fence-free kernels -> {GOLDEN}")
    else:
        for k, v in c.items():
            if all(a in k for a in args):
                print(f"{v['stores']:4d} stores  {v['plain']:3d} plain  {v['atomics']:3d} atomics  {k}")
                for ln in v.get("plain_lines", []):
                    print("        " + ln)
