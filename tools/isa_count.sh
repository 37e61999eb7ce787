#!/bin/bash
# static instruction counts per kernel of a built library:  tools/isa_count.sh <lib.so> <kernel-substring>
d=$(mktemp -d); cp $1 $d/lib.so; cd $d; /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null 2>&1
for co in *gfx950*; do /opt/rocm/lib/llvm/bin/llvm-objdump -d $co > $co.s 2>/dev/null; done
python3 - "$2" <<'PY'
import sys, re, glob, subprocess
for fn in glob.glob('*.s'):
    txt = open(fn).read()
    for f in re.split(r'\n(?=[0-9a-f]+ <[^>]+>:)', txt):
        m = re.match(r'[0-9a-f]+ <([^>]+)>:', f)
        if not m: continue
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        if sys.argv[1] not in name: continue
        c = lambda p: len(re.findall(p, f))
        pats = [("valu", r"\tv_"), ("f64", r"v_\w+_f64"), ("salu", r"\ts_"), ("s_load", r"\ts_load"), ("waitcnt", "s_waitcnt"), ("ds", r"\tds_"),
                ("vmem", r"\tglobal_"), ("branch", "s_cbranch"), ("readlane", "v_readlane"), ("writelane", "v_writelane")]
        print(f"{name[:44]:44s} " + " ".join(f"{k} {c(p):4d}" for k, p in pats))
PY
rm -rf $d
