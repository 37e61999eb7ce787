"""Throw-away helper process of eryn_amd.ladder.LadderPipeline: ``python -m eryn_amd.pipe_probe <device> <rank>
<nranks> <dir> <timeout_s>`` runs hens_pipe_selftest and exits 0 on success (no torch import: it must start fast
and must not share a HIP runtime with anything)."""
import sys

from . import _lib


def main(argv):
    device, rank, nranks = int(argv[1]), int(argv[2]), int(argv[3])
    code = _lib.load().hens_pipe_selftest(device, rank, nranks, argv[4].encode(), float(argv[5]))
    if code != _lib.HENS_OK:
        msg = _lib.load().hens_last_error(None)
        print(f"pipe_probe rank {rank}: {msg.decode() if msg else code}", flush=True)
    return 0 if code == _lib.HENS_OK else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))
