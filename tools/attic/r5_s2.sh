#!/bin/bash
# Round 5, session 2: same-box A/B of the pipeline rank (base = round 4's library) + wall-clock phase stamps of both
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5_s2; mkdir -p $out
AB=$GRAFT_REPO_ROOT/ab_live
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
{
for L in base_rt new_rt; do for d in 0 1; do
  echo "=== $L delay $d"; HENS_LIB=$AB/libhens_$L.so PIPE_DELAY=$d timeout 200 python tools/pipe_trace.py 8 16384 64 2>&1 | grep -v "amdgpu.ids\|^  repeat [0-3]"
  HENS_LIB=$AB/libhens_$L.so PIPE_DELAY=$d timeout 200 python tools/pipe_trace.py 16 4096 32 2>&1 | grep -v "amdgpu.ids\|^  repeat [0-3]"
done; done
echo "=== new_rt single"; HENS_LIB=$AB/libhens_new_rt.so timeout 200 python tools/pipe_trace.py 8 16384 64 single 2>&1 | grep -v "amdgpu.ids\|^  repeat [0-3]"
HENS_LIB=$AB/libhens_new_rt.so timeout 200 python tools/pipe_trace.py 16 4096 32 single 2>&1 | grep -v "amdgpu.ids\|^  repeat [0-3]"
} > $out/pipe_trace.txt 2>&1
cat $out/pipe_rank_ab.txt | cut -c1-170; grep -v Traceback $out/pipe_trace.txt | cut -c1-200 | head -150
