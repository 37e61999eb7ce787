#!/bin/bash
# Round 5, session 1: the pipeline rank's first launch before / after (count wait on the adapting wave alone, chain split around the
# first barrier, flag waits on the shader clock) - parity, per-launch timings (three alternations), wall-clock phase stamps.
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5_s1; mkdir -p $out
AB=$GRAFT_REPO_ROOT/ab_live
timeout 1500 python -m pytest tests/test_hip_pipeline.py tests/test_hip_repeat.py -x -q -m gpu > $out/pytest_pipeline.txt 2>&1; tail -3 $out/pytest_pipeline.txt
{
for rep in 1 2 3; do for L in base new; do
  if [ $L = base ]; then export HENS_LIB=$AB/libhens_base.so; else unset HENS_LIB; fi
  for d in 0 1; do
    echo -n "[$L] "; PIPE_DELAY=$d timeout 200 python tools/pipe_prof.py 8 16384 64 200 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
    echo -n "[$L] "; PIPE_DELAY=$d timeout 200 python tools/pipe_prof.py 16 4096 32 400 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
  done
done; done
} > $out/pipe_rank_ab.txt 2>&1
{
for L in base_rt new_rt; do for d in 0 1; do
  echo "=== $L delay $d"; HENS_LIB=$AB/libhens_$L.so PIPE_DELAY=$d timeout 200 python tools/pipe_trace.py 8 16384 64 2>&1 | grep -v amdgpu.ids
  HENS_LIB=$AB/libhens_$L.so PIPE_DELAY=$d timeout 200 python tools/pipe_trace.py 16 4096 32 2>&1 | grep -v amdgpu.ids
done; done
echo "=== new_rt single"; HENS_LIB=$AB/libhens_new_rt.so timeout 200 python tools/pipe_trace.py 8 16384 64 single 2>&1 | grep -v amdgpu.ids
HENS_LIB=$AB/libhens_new_rt.so timeout 200 python tools/pipe_trace.py 16 4096 32 single 2>&1 | grep -v amdgpu.ids
} > $out/pipe_trace.txt 2>&1
{
echo "=== shader-clock stamps, main library"; for d in 0 1; do PIPE_DELAY=$d timeout 200 python tools/pipe_trace.py 8 16384 64 2>&1 | grep -v amdgpu.ids; done
timeout 200 python tools/trace_phases.py 8 16384 64 2>&1 | grep -v amdgpu.ids
timeout 200 python tools/trace_phases.py 16 4096 32 2>&1 | grep -v amdgpu.ids
} > $out/trace_cycles.txt 2>&1
{
for rep in 1 2 3; do
  echo -n "flush default: "; python bench.py --steps 20 --warmup 5 --no-cpu --no-other 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step']*1e3, d['value'])"
  echo -n "HENS_AQL_FLUSH=2: "; HENS_AQL_FLUSH=2 python bench.py --steps 20 --warmup 5 --no-cpu --no-other 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step']*1e3, d['value'])"
done
} > $out/flush_ab.txt 2>&1
cat $out/pipe_rank_ab.txt; cat $out/flush_ab.txt; tail -60 $out/pipe_trace.txt
