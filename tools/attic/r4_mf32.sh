#!/bin/bash
export PYTHONPATH=$PWD
for L in d32 d32mf; do
  export HENS_LIB=$PWD/build_ab/libhens_$L.so
  echo "== $L"; python tools/like_check.py 16 1024 32 2>&1 | tail -2; python tools/like_check.py 4 500 32 nonsym 2>&1 | tail -1
done
for rep in 1 2 3; do for L in d32 d32mf; do
  export HENS_LIB=$PWD/build_ab/libhens_$L.so
  echo -n "$L: "; python tools/short_call.py 2000 2>&1 | tail -1
done; done
for L in d32 d32mf; do export HENS_LIB=$PWD/build_ab/libhens_$L.so; echo -n "$L K=20: "; python tools/short_call.py 20 2>&1 | tail -1; done
