#!/bin/bash
# round 6, session 2: dispatch-timestamp test again; queue profiling on at creation vs late (timing A/B); stagger sweep
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06b; mkdir -p $out; cd $R; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_hip_records.py -m gpu -x -q -k "dispatch_timestamps" > $out/pytest_new.txt 2>&1; tail -5 $out/pytest_new.txt
for rep in 1 2; do
  for late in 0 1; do
    if [ $late = 1 ]; then export HENS_AQL_PROF_LATE=1; else unset HENS_AQL_PROF_LATE; fi
    echo -n "prof_late=$late: "; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-other 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step']*1e3, [k['avg_launch_us'] for k in r['kernels']], r['sum_kernel_us_per_iteration'], r['span_us_per_iteration'])"
  done
done 2>&1 | tee $out/prof_late_ab.txt
unset HENS_AQL_PROF_LATE
timeout 900 python tools/stagger_sweep.py 8 16384 64 dense 0,16,32,48,64,80,96,128 3 2 2>&1 | grep -v amdgpu.ids | tee $out/stagger_cfg3.txt
timeout 600 python tools/stagger_sweep.py 8 16384 64 dense 0,1073741856,1073741888,1073741920 3 1 2>&1 | grep -v amdgpu.ids | tee -a $out/stagger_cfg3.txt
timeout 600 python tools/stagger_sweep.py 8 16384 64 dense 32,64,96 1 1 2>&1 | grep -v amdgpu.ids | tee -a $out/stagger_cfg3.txt
timeout 600 python tools/stagger_sweep.py 8 16384 64 dense 32,64,96 2 1 2>&1 | grep -v amdgpu.ids | tee -a $out/stagger_cfg3.txt
timeout 600 python tools/stagger_sweep.py 4 8192 128 rosen 0,32,64,96 3 1 2>&1 | grep -v amdgpu.ids | tee $out/stagger_cfg5.txt
