#!/bin/bash
# round 6, session 27: bench.py --gpus 2 (two processes sharing this GPU, gloo for the host-side collectives) on a D = 64 shape with the
# persistent first launch forced: run_sharded + k_stretch2<PIPE> + HIP-IPC mailboxes in one go
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06v; mkdir -p $out; cd $R; export PYTHONPATH=$R
HENS_TILE2_FORCE=1 HENS_TILE2_LOG=1 HENS_DIST_BACKEND=gloo GPU_MAX_HW_QUEUES=16 timeout 600 python bench.py --gpus 2 --ntemps 8 --nwalkers 1024 --ndim 64 --steps 20 --warmup 5 --no-cpu > $out/bench_gpus2_d64.json 2> $out/err.log
echo "rc $?"; grep -c "k_stretch2<pipe=1>" $out/err.log; head -c 900 $out/bench_gpus2_d64.json; echo; grep -v "amdgpu.ids\|socket.cpp\|k_stretch2" $out/err.log | tail -5
