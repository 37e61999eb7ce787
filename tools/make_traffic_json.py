"""profiles/traffic.json from the rocprofv3 --pmc passes written by tools/profile_bench.sh.

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 reports KB).  The factor 2 is the gfx950 FETCH_SIZE
correction of MI355X_MICROARCH.md (wide coalesced reads are tallied at half their bytes); it is calibrated in the same
run on the eval launch (k_stretch_fast<.., MODE_EVAL>), which streams a known 16.78 MB once.
bench.py reads the per-kernel totals ("k_stretch_fast", "k_split1_pt") as a static, labelled figure.

    python tools/make_traffic_json.py <tag> [dir]      # reads <dir>/<tag>_pmc_{FETCH,WRITE}_SIZE.txt
"""
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = sys.argv[2] if len(sys.argv) > 2 else "profiles"


def mean_of(path, pattern, required=True):
    for line in open(path):
        if re.search(pattern, line):
            return float(re.search(r"mean\s+([0-9.]+)", line).group(1))
    if required:
        raise SystemExit(f"{pattern} not found in {path}")
    return None


F, Wf = f"{src}/{tag}_pmc_FETCH_SIZE.txt", f"{src}/{tag}_pmc_WRITE_SIZE.txt"
cal_fetch = mean_of(F, r"k_stretch_fast<32, 0, 1,")
cal_write = mean_of(Wf, r"k_stretch_fast<32, 0, 1,")
known_read = 16 * 4096 * 32 * 8          # eval launch: every row once
known_write = 2 * 16 * 4096 * 8          # eval launch: logl + logp
out = {
    "shape": [16, 4096, 32],                 # (ntemps, nwalkers, ndim) of the profiled command: bench.py's default
    "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), profiles/{tag}_pmc_*.txt",
    "fetch_correction": 2.0,
    "calibration": {"eval_kernel_fetch_kb_reported": cal_fetch, "eval_kernel_known_read_bytes": known_read,
                    "reported_over_known": cal_fetch * 1024 / known_read,
                    "eval_kernel_write_kb_reported": cal_write, "eval_kernel_known_write_bytes": known_write},
    "detail": {},
}
for name, pat in (("k_stretch_fast", r"k_stretch_fast<32, 0, 0,"), ("k_split1_pt", r"k_split1_pt<32, 0,"),
                  ("k_pt_cascade", r"k_pt_cascade<true>")):
    f, w = mean_of(F, pat, False), mean_of(Wf, pat, False)
    if f is None or w is None:
        continue
    out[name] = (2.0 * f + w) * 1024
    out["detail"][name] = {"fetch_size_kb_reported": f, "write_size_kb_reported": w}
json.dump(out, open(f"{src}/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
