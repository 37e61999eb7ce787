#!/bin/bash
# round 6, session 22: the r06z bench lines again with the final bench.py (launch_stat, fit_tolerance, profiled_over_timed)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06z; mkdir -p $out; cd $R; export PYTHONPATH=$R
python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench.err
python bench.py --no-cpu --no-other > $out/bench.json 2>> $out/bench.err
python bench.py --workload cfg4 > $out/bench_cfg4.json 2>> $out/bench.err
python bench.py --workload cfg5 > $out/bench_cfg5.json 2>> $out/bench.err
HENS_DIST_BACKEND=gloo GPU_MAX_HW_QUEUES=16 python bench.py --gpus 2 --ntemps 8 --nwalkers 256 --ndim 32 --steps 20 --warmup 5 --no-cpu > $out/bench_gpus2_dryrun.json 2>> $out/bench.err
python -m pytest tests/test_hip_bench_multi.py tests/test_hip_sampler.py -q -x 2>&1 | tail -3
head -c 400 $out/bench_steps20.json; tail -5 $out/bench.err
