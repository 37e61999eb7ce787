"""bench.py --gpus N end to end as a DRY RUN on one GPU (-m gpu): two ranks over gloo share the device (RCCL refuses that),
small shards.  Not a measurement: it pins that the N > 1 line carries everything the scaling record needs - the one-sided
pipeline on both adaptation schedules, the RCCL neighbour transport, the weak-scaling base, an efficiency for each, the
world size the backend saw and the per-rank wait breakdown."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_dry_run_has_every_key():
    env = dict(os.environ, HENS_DIST_BACKEND="gloo", GPU_MAX_HW_QUEUES="16", HENS_PIPE_TIMEOUT_S="20",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ntemps", "8", "--nwalkers", "256",
                        "--ndim", "32", "--steps", "6", "--warmup", "4", "--no-cpu"], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    cfg = out["config"]
    assert cfg["world_size_seen_by_backend"] == 2 and cfg["dist_backend"] == "gloo" and "one-sided" in cfg["transport"]
    assert out["weak_base"]["ms_per_step"] > 0
    for leg in (out, out["delayed_adaptation"], out["rccl_neighbour"]):
        assert leg["ms_per_step"] > 0 and 0 < leg["efficiency"] < 10
        assert abs(leg["efficiency"] - out["weak_base"]["ms_per_step"] / leg["ms_per_step"]) < 1e-12
    assert "ncclSend" in out["rccl_neighbour"]["transport"]
    waits = out["rank_waits"]
    for sched in ("adaptation_delay_0", "adaptation_delay_1"):
        assert set(waits[sched]) == {"rank0", "rank1"}
        for rk in waits[sched].values():
            assert "error" not in rk and "walk:columns" in rk and "stretch:counts" in rk
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] <= 1.0
