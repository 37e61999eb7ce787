#!/bin/bash
# Round-6 evidence: the contract bench (the driver's 20-step blocks and long blocks), rocprofv3 kernel trace + FETCH / WRITE passes at
# config 2, one config-3 shard, ALL of config 5 on one GPU and one config-5 shard, the SQ pass, the RJ profile with the instruction
# classes, config 4 / 5 lines, the pipeline rank (lead workgroup without a tile / beside its tile), the 2-rank dry run.
#     bash tools/r6_final_profiles.sh <tag>
tag=${1:-r06z}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
export PYTHONPATH=$R
cd $R
python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench.err
python bench.py --no-cpu --no-other > $out/bench.json 2>> $out/bench.err
bash tools/profile_bench.sh $tag > $out/profile.log 2>&1
bash tools/pmc_sq.sh 16 4096 32 > $out/pmc_sq.txt 2>&1
bash tools/profile_bench.sh ${tag}_cfg3shard --ntemps 8 --nwalkers 16384 --ndim 64 > $out/profile_cfg3.log 2>&1
bash tools/profile_bench.sh ${tag}_cfg5 --workload cfg5 > $out/profile_cfg5.log 2>&1
bash tools/profile_bench.sh ${tag}_cfg5shard --workload cfg5 --ntemps 4 > $out/profile_cfg5shard.log 2>&1
cd $R
python bench.py --workload cfg4 > $out/bench_cfg4.json 2>> $out/bench.err
python bench.py --workload cfg5 > $out/bench_cfg5.json 2>> $out/bench.err
bash tools/profile_rj.sh ${tag}_cfg4 > $out/profile_rj.log 2>&1
cd $R
{ for noad in 0; do
    for d in 0 1; do PIPE_DELAY=$d python tools/pipe_prof.py 8 16384 64 200; PIPE_DELAY=$d python tools/pipe_prof.py 16 4096 32 400; PIPE_DELAY=$d python tools/pipe_prof.py 4 8192 128 200; done
  done; } 2>&1 | grep -v amdgpu.ids > $out/pipe_rank.txt
HENS_DIST_BACKEND=gloo GPU_MAX_HW_QUEUES=16 python bench.py --gpus 2 --ntemps 8 --nwalkers 256 --ndim 32 --steps 20 --warmup 5 --no-cpu > $out/bench_gpus2_dryrun.json 2>> $out/bench.err
ls $out $R/gpurun_out/${tag}_cfg3shard $R/gpurun_out/${tag}_cfg5 $R/gpurun_out/${tag}_cfg5shard $R/gpurun_out/${tag}_cfg4 | head -60
head -c 300 $out/bench_steps20.json; echo; cat $out/pipe_rank.txt | cut -c1-130
