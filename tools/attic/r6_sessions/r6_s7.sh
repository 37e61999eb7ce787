#!/bin/bash
# round 6, session 7: the pipeline rank's lead workgroup without a tile (StretchArgs::adapter): tests, then A/B against workgroup (0,0)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06g; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 1500 python -m pytest tests/test_hip_pipeline.py tests/test_hip_fullsize.py tests/test_hip_sharded.py -m gpu -x -q > $out/pytest_pipe.txt 2>&1; tail -6 $out/pytest_pipe.txt
for rep in 1 2 3; do
  for off in 1 0; do
    if [ $off = 1 ]; then export HENS_PIPE_NO_ADAPTER=1; tag="wg(0,0)"; else unset HENS_PIPE_NO_ADAPTER; tag="adapter"; fi
    for shape in "8 16384 64 300" "16 4096 32 600" "4 8192 128 300"; do
      echo -n "[$tag] "; timeout 300 python tools/pipe_prof.py $shape 2>&1 | grep -v amdgpu.ids | grep "^pipe"
    done
  done
done 2>&1 | tee $out/adapter_ab.txt
unset HENS_PIPE_NO_ADAPTER
timeout 300 python tools/pipe_prof.py 8 16384 64 300 2>&1 | grep "^single" | tee -a $out/adapter_ab.txt
timeout 300 python tools/pipe_prof.py 16 4096 32 600 2>&1 | grep "^single" | tee -a $out/adapter_ab.txt
