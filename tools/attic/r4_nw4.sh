#!/bin/bash
export PYTHONPATH=$PWD
for rep in 1 2; do for L in main nw4; do
  if [ $L = main ]; then unset HENS_LIB; else export HENS_LIB=$PWD/build_ab/libhens_$L.so; fi
  echo -n "$L diag: "; python tools/quick_bench.py --T 8 --W 16384 --D 64 --like diag --steps 1000 2>&1 | head -1 | cut -c40-
done; done
for L in main nw4; do
  if [ $L = main ]; then unset HENS_LIB; else export HENS_LIB=$PWD/build_ab/libhens_$L.so; fi
  echo -n "$L diag: "; python tools/quick_bench.py --T 8 --W 16384 --D 64 --like diag --steps 400 --prof 1 2>&1 | sed -n 3,3p
  echo -n "$L rosen: "; python tools/quick_bench.py --T 8 --W 16384 --D 64 --like rosen --steps 400 --prof 1 2>&1 | sed -n 3,3p
done
