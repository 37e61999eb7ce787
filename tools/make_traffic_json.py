"""profiles/traffic.json from the rocprofv3 --pmc passes written by tools/profile_bench.sh.

HBM bytes per stretch launch = 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 reports KB).  The factor 2 is the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (wide coalesced reads are tallied at half their
bytes); it is calibrated in the same run on the eval kernel, which streams a known 16.78 MB once.
"""
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else f"profiles"


def mean_of(path, pattern):
    for line in open(path):
        if re.search(pattern, line):
            return float(re.search(r"mean\s+([0-9.]+)", line).group(1))
    raise SystemExit(f"{pattern} not found in {path}")


fetch = mean_of(f"{src}/{tag}_pmc_FETCH_SIZE.txt", r"k_stretch_fast<32, 0, 0,")
write = mean_of(f"{src}/{tag}_pmc_WRITE_SIZE.txt", r"k_stretch_fast<32, 0, 0,")
cal_fetch = mean_of(f"{src}/{tag}_pmc_FETCH_SIZE.txt", r"k_stretch_fast<32, 0, 1,")
cal_write = mean_of(f"{src}/{tag}_pmc_WRITE_SIZE.txt", r"k_stretch_fast<32, 0, 1,")
known_read = 16 * 4096 * 32 * 8          # eval kernel: every row once
known_write = 2 * 16 * 4096 * 8          # eval kernel: logl + logp
out = {
    "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), profiles/{tag}_pmc_*.txt",
    "kernel": "k_stretch_fast<32, dense, step, NW=8>",
    "fetch_size_kb_reported": fetch, "write_size_kb_reported": write,
    "fetch_correction": 2.0,
    "calibration": {"eval_kernel_fetch_kb_reported": cal_fetch, "eval_kernel_known_read_bytes": known_read,
                    "reported_over_known": cal_fetch * 1024 / known_read,
                    "eval_kernel_write_kb_reported": cal_write, "eval_kernel_known_write_bytes": known_write},
    "stretch_bytes_per_launch": (2.0 * fetch + write) * 1024,
    "algorithmic_bytes_per_launch": (24 * 32 + 32) * 16 * 4096 / 2,
}
json.dump(out, open(f"{src}/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
