#!/bin/bash
# fused in-place iteration on pipeline ranks: correctness (pipeline / fullsize tests) and same-box timing against the three
# copying launches (HENS_PIPE_NO_FUSED=1)
out=gpurun_out/${1:-s3d}; mkdir -p $out
export PYTHONPATH=. GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_hip_pipeline.py tests/test_hip_fullsize.py::test_config3_ladder_single_context_equals_8_shard_pipeline -x -q > $out/pytest_pipe.log 2>&1
tail -15 $out/pytest_pipe.log
{
for d in 0 1; do PIPE_DELAY=$d python tools/pipe_prof.py 8 16384 64 200; PIPE_DELAY=$d python tools/pipe_prof.py 16 4096 32 400 | head -1; done
for knob in "X=1" "HENS_PIPE_NO_FUSED=1"; do
  echo "== $knob"
  env $knob HENS_PIPE_STATS=1 timeout 300 python tools/time_pipeline.py local 2 16 4096 32 400
  env $knob PIPE_DELAY=1 timeout 300 python tools/time_pipeline.py local 2 16 4096 32 400
done
} 2>&1 | grep -v amdgpu.ids > $out/time_pipe.log
cat $out/time_pipe.log
