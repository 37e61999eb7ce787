#!/bin/bash
# round 6, session 29: the whole -m gpu suite on the final tree
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06z; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
