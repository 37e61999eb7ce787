#!/bin/bash
# round 6, session 12: k_stretch2 with 0 / 1 / 2 gather passes of the next tile in front of the likelihood phase
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r06q; mkdir -p $out; cd $R; export PYTHONPATH=$R
for rep in 1 2; do
  for l in main prec1 prec2; do
    if [ $l = main ]; then unset HENS_LIB; else export HENS_LIB=$R/ab_live/libhens_$l.so; fi
    for shape in "8 16384 64 dense" "8 16384 64 diag" "16 16384 64 dense"; do echo -n "[$l] "; timeout 600 python tools/tile2_check.py $shape 200 2>&1 | grep -v amdgpu.ids; done
  done
done | tee $out/prec_ab.txt
