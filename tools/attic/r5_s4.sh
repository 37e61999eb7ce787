#!/bin/bash
# Round 5, session 4: the count push in one round trip on a wave of its own, own sums out of registers
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5_s6; mkdir -p $out
AB=$GRAFT_REPO_ROOT/ab_live
timeout 900 python -m pytest tests/test_hip_pipeline.py tests/test_hip_fullsize.py tests/test_hip_replay.py -x -q -m gpu 2>&1 | tail -3 > $out/pytest.txt; cat $out/pytest.txt
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
cut -c1-150 $out/pipe_rank_ab.txt | sed 's/stretch launch/S/; s/fused launch/F/; s/, cascade.*//' | grep pipe
{
for L in new_rt; do for d in 0 1; do
  echo "=== $L delay $d"; HENS_LIB=$AB/libhens_$L.so PIPE_DELAY=$d timeout 200 python tools/pipe_trace.py 8 16384 64 2>&1 | grep -v "amdgpu.ids\|^  repeat [0-3]"
  HENS_LIB=$AB/libhens_$L.so PIPE_DELAY=$d timeout 200 python tools/pipe_trace.py 16 4096 32 2>&1 | grep -v "amdgpu.ids\|^  repeat [0-3]"
done; done
} > $out/pipe_trace.txt 2>&1
grep -A3 "workgroup (0,0):" $out/pipe_trace.txt | grep -v "p90" | cut -c1-120
bash tools/pipe_slack.sh 8 16384 64 inj64 > $out/slack64.txt 2>&1
bash tools/pipe_slack.sh 16 4096 32 inj32 400 > $out/slack32.txt 2>&1
unset HENS_LIB
cat $out/slack64.txt $out/slack32.txt | cut -c1-140

