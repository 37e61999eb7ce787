#!/bin/bash
# Round 5, last session: soaks of the paths this session touched (RJ repeats and fold equivalence, single-GPU equivalences, pipeline)
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R; out=gpurun_out/r5_s19; mkdir -p $out
{ timeout 600 python tools/soak_rj.py 3000 3 2>&1 | grep -v amdgpu | tail -6; } | cut -c1-200 > $out/soak_rj.txt; cat $out/soak_rj.txt
{ timeout 600 python tools/soak_rj_fold.py 6000 2>&1 | grep -v amdgpu | tail -4; } | cut -c1-200 > $out/soak_rj_fold.txt; cat $out/soak_rj_fold.txt
{ timeout 900 python tools/soak_equivalence.py 2>&1 | grep -v amdgpu | tail -14; } | cut -c1-200 > $out/soak_equiv.txt; cat $out/soak_equiv.txt
{ timeout 900 python tools/soak_pipeline.py 2>&1 | grep -v amdgpu | tail -12; } | cut -c1-200 > $out/soak_pipeline.txt; cat $out/soak_pipeline.txt
