#!/bin/bash
# Round 5, session 3: steady-state wall-clock stamps of the pipeline rank's first launch (base / new), A/B timings, injection tables
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5_s3; mkdir -p $out
AB=$GRAFT_REPO_ROOT/ab_live
{
for L in base_rt new_rt; do for d in 0 1; do
  echo "=== $L delay $d"; HENS_LIB=$AB/libhens_$L.so PIPE_DELAY=$d timeout 200 python tools/pipe_trace.py 8 16384 64 2>&1 | grep -v "amdgpu.ids\|^  repeat [0-3]"
  HENS_LIB=$AB/libhens_$L.so PIPE_DELAY=$d timeout 200 python tools/pipe_trace.py 16 4096 32 2>&1 | grep -v "amdgpu.ids\|^  repeat [0-3]"
done; done
} > $out/pipe_trace.txt 2>&1
{
for rep in 1 2; do for L in base new; do
  if [ $L = base ]; then export HENS_LIB=$AB/libhens_base.so; else unset HENS_LIB; fi
  for d in 0 1; do
    PIPE_DELAY=$d timeout 200 python tools/pipe_prof.py 8 16384 64 200 2>&1 | grep -E "^pipe|^single" | sed "s/^/[$L] /"
    PIPE_DELAY=$d timeout 200 python tools/pipe_prof.py 16 4096 32 400 2>&1 | grep -E "^pipe|^single" | sed "s/^/[$L] /"
  done
done; done
} > $out/pipe_rank_ab.txt 2>&1
unset HENS_LIB
bash tools/pipe_slack.sh 8 16384 64 inj64 > $out/slack64.txt 2>&1
bash tools/pipe_slack.sh 16 4096 32 inj32 400 > $out/slack32.txt 2>&1
unset HENS_LIB
timeout 900 python -m pytest tests/test_hip_pipeline.py -x -q -m gpu 2>&1 | tail -2 > $out/pytest.txt
cat $out/pytest.txt; cut -c1-150 $out/pipe_rank_ab.txt | sed 's/stretch launch/S/; s/fused launch/F/; s/, cascade.*//'; cat $out/slack64.txt $out/slack32.txt | cut -c1-150
