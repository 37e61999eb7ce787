"""VGPR / SGPR / spill / LDS figures of the built kernels, from the code object's metadata notes.
  python tools/kernel_regs.py [substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("HENS_LIB") or os.path.join(ROOT, "eryn_amd", "lib", "libhipensemble.so")
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    so = os.path.join(d, "lib.so")
    os.symlink(LIB, so)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], cwd=d, capture_output=True)
    # (one code object per translation unit since round 4: all of them)
    txt = "".join(subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(d, co)], capture_output=True, text=True).stdout
                  for co in sorted(f for f in os.listdir(d) if "gfx950" in f))
rows = []
for b in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
    g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", b) or [None, "0"])[1]
    rows.append((g("name"), int(g("vgpr_count")), int(g("vgpr_spill_count")), int(g("sgpr_count")), int(g("private_segment_fixed_size"))))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
for (n, v, sp, sg, ps), dn in zip(rows, names):
    if not sys.argv[1:] or all(s in dn for s in sys.argv[1:]):
        print(f"{v:4d} vgpr  spill {sp:3d}  scratch {ps:5d} B  sgpr {sg:3d}  {dn[:120]}")
