#!/bin/bash
export PYTHONPATH=$PWD
for L in head main; do
  if [ $L = main ]; then unset HENS_LIB; else export HENS_LIB=$PWD/build_ab/libhens_$L.so; fi
  echo "== $L"; python tools/like_check.py 4 2048 128 2>&1 | tail -2; python tools/like_check.py 3 500 128 nonsym 2>&1 | tail -2
  for s in "4 8192" "8 8192"; do python tools/quick_bench.py --T ${s%% *} --W 8192 --D 128 --steps 500 --prof 0 2>&1 | head -1 | cut -c1-120; done
  python tools/quick_bench.py --T 4 --W 8192 --D 128 --steps 300 --prof 1 2>&1 | sed -n 3,3p
done
