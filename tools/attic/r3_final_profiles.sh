#!/bin/bash
# Round-3 end state: the contract bench (long blocks and the driver's 20-step blocks), rocprofv3 kernel trace + FETCH / WRITE and
# SQ counter passes at config 2, the same at one config-3 shard, and the pipeline rank's per-launch times.
tag=${1:-r03c}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
export PYTHONPATH=$R
cd $R
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu > $out/bench_steps20.json 2>> $out/bench.err
bash tools/profile_bench.sh $tag > $out/profile.log 2>&1
bash tools/pmc_sq.sh 16 4096 32 > $out/pmc_sq.txt 2>&1
bash tools/profile_bench.sh ${tag}_cfg3shard --ntemps 8 --nwalkers 16384 --ndim 64 > $out/profile_cfg3.log 2>&1
{ for d in 0 1; do PIPE_DELAY=$d python tools/pipe_prof.py 8 16384 64 200; done; PIPE_DELAY=1 python tools/pipe_prof.py 16 4096 32 400; } 2>&1 | grep -v amdgpu.ids > $out/pipe_rank.txt
HENS_PIPE_STATS=1 GPU_MAX_HW_QUEUES=16 python tools/time_pipeline.py local 2 16 4096 32 400 2>&1 | grep -v amdgpu.ids >> $out/pipe_rank.txt
PIPE_DELAY=1 GPU_MAX_HW_QUEUES=16 python tools/time_pipeline.py local 2 16 4096 32 400 2>&1 | grep -v amdgpu.ids >> $out/pipe_rank.txt
HENS_DIST_BACKEND=gloo GPU_MAX_HW_QUEUES=16 python bench.py --gpus 2 --ntemps 8 --nwalkers 256 --ndim 32 --steps 20 --warmup 5 --no-cpu > $out/bench_gpus2_dryrun.json 2>> $out/bench.err
python bench.py --workload cfg4 > $out/bench_cfg4.json 2>> $out/bench.err
python bench.py --workload cfg5 > $out/bench_cfg5.json 2>> $out/bench.err
ls -la $out $R/gpurun_out/${tag}_cfg3shard
head -c 600 $out/bench.json; echo; head -c 300 $out/bench_steps20.json; echo; cat $out/pipe_rank.txt
