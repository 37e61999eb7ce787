#!/bin/bash
# round 6, session 5: lazy drop-in timing after the record-mode report; NW = 4 at D = 64 (one round of 4-wave workgroups) with the diagonal likelihood
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06e; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_hip_sampler.py -m gpu -x -q -s -k "device_draw or lazy_device" > $out/pytest_lazy.txt 2>&1; tail -8 $out/pytest_lazy.txt
for rep in 1 2; do
  unset HENS_LIB; timeout 300 python tools/stagger_sweep.py 8 16384 64 diag 0 3 1 2>&1 | grep -v amdgpu.ids | sed 's/^/NW8 /'
  HENS_LIB=$R/ab_live/libhens_nw4.so timeout 300 python tools/stagger_sweep.py 8 16384 64 diag 0 3 1 2>&1 | grep -v amdgpu.ids | sed 's/^/NW4 /'
done | tee $out/nw4_diag.txt
for rep in 1; do
  unset HENS_LIB; timeout 300 python tools/stagger_sweep.py 16 16384 64 diag 0 3 1 2>&1 | grep -v amdgpu.ids | sed 's/^/NW8 /'
  HENS_LIB=$R/ab_live/libhens_nw4.so timeout 300 python tools/stagger_sweep.py 16 16384 64 diag 0 3 1 2>&1 | grep -v amdgpu.ids | sed 's/^/NW4 /'
  unset HENS_LIB; timeout 300 python tools/stagger_sweep.py 8 4096 64 diag 0 3 1 2>&1 | grep -v amdgpu.ids | sed 's/^/NW8 /'
  HENS_LIB=$R/ab_live/libhens_nw4.so timeout 300 python tools/stagger_sweep.py 8 4096 64 diag 0 3 1 2>&1 | grep -v amdgpu.ids | sed 's/^/NW4 /'
done | tee -a $out/nw4_diag.txt
