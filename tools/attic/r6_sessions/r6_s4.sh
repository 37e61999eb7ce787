#!/bin/bash
# round 6, session 4: device-draw drop-in moves + lazy State on the GPU
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06d; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_hip_sampler.py -m gpu -x -q -s -k "device_draw or lazy_device" > $out/pytest_lazy.txt 2>&1; tail -25 $out/pytest_lazy.txt
timeout 900 python -m pytest tests/test_hip_sampler.py tests/test_hip_rj.py tests/test_hip_rng.py -m gpu -x -q > $out/pytest_more.txt 2>&1; tail -5 $out/pytest_more.txt
