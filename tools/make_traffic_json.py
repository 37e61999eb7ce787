"""profiles/traffic.json from the rocprofv3 --pmc passes written by tools/profile_bench.sh, one entry per profiled shape.

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 reports KB).  The factor 2 is the gfx950 FETCH_SIZE
correction of MI355X_MICROARCH.md (wide coalesced reads are tallied at half their bytes); it is calibrated in the same
run on the eval launch (k_stretch_fast<.., MODE_EVAL>), which streams every row of the state exactly once.
bench.py reads the per-kernel totals of its shape as a static, labelled figure (roofline.frac_traffic).

    python tools/make_traffic_json.py <tag> T W D [like] [dir]    # reads <dir>/<tag>_pmc_{FETCH,WRITE}_SIZE.txt
"""
import json
import os
import re
import sys

tag = sys.argv[1]
T, W, D = (int(v) for v in sys.argv[2:5])
like = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # 0 dense Gaussian, 1 diagonal, 2 Rosenbrock
src = sys.argv[6] if len(sys.argv) > 6 else "profiles"


def mean_of(path, pattern):
    for line in open(path):
        if re.search(pattern, line):
            return float(re.search(r"mean\s+([0-9.]+)", line).group(1))
    return None


F, Wf = f"{src}/{tag}_pmc_FETCH_SIZE.txt", f"{src}/{tag}_pmc_WRITE_SIZE.txt"
cal_fetch = mean_of(F, rf"k_stretch_fast<{D}, {like}, 1,")
cal_write = mean_of(Wf, rf"k_stretch_fast<{D}, {like}, 1,")
known_read = T * W * D * 8               # eval launch: every row once
entry = {
    "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), profiles/{tag}_pmc_*.txt",
    "fetch_correction": 2.0,
    "calibration": None if cal_fetch is None else {
        "eval_kernel_fetch_kb_reported": cal_fetch, "eval_kernel_known_read_bytes": known_read,
        "reported_over_known": cal_fetch * 1024 / known_read, "eval_kernel_write_kb_reported": cal_write},
    "detail": {},
}
for name, pat in (("k_stretch_fast", rf"k_stretch_fast<{D}, {like}, 0,"), ("k_stretch_fast_mh", rf"k_stretch_fast<{D}, {like}, 2,"),
                  ("k_split1_pt", rf"k_split1_pt<{D}, {like},"), ("k_iter", rf"k_iter<{D}, {like},"), ("PT", r"k_pt_cascade<true>")):
    f, w = mean_of(F, pat), mean_of(Wf, pat)
    if (f is None or w is None) and name == "k_stretch_fast":          # launches of more than one round: the persistent kernel (hens_tile2.h)
        f, w = mean_of(F, rf"k_stretch2<{D}, {like}[,>]"), mean_of(Wf, rf"k_stretch2<{D}, {like}[,>]")
    if f is None or w is None:
        continue
    entry[name] = (2.0 * f + w) * 1024
    entry["detail"][name] = {"fetch_size_kb_reported": f, "write_size_kb_reported": w}
path = f"{src}/traffic.json"
doc = json.load(open(path)) if os.path.exists(path) else {}
if "shapes" not in doc:
    doc = {"shapes": {}}
doc["shapes"][f"{T}x{W}x{D}"] = entry
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps({k: v for k, v in entry.items() if k != "detail"}, indent=1))
