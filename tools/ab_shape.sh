#!/bin/bash
# Same-box A/B of dev builds at one shape: tools/ab_shape.sh T W D lib1 lib2 ...  (libs under ab_live/; "main" = the in-tree library)
export PYTHONPATH=$GRAFT_REPO_ROOT
T=$1; W=$2; D=$3; shift 3
for rep in 1 2 3; do
  for L in "$@"; do
    if [ $L = main ]; then unset HENS_LIB; else export HENS_LIB=$GRAFT_REPO_ROOT/ab_live/libhens_$L.so; fi
    echo -n "$L: "; timeout 200 python tools/quick_bench.py --T $T --W $W --D $D --steps 1000 --prof 0 2>&1 | head -1 | cut -c40-
  done
done
for L in "$@"; do
  if [ $L = main ]; then unset HENS_LIB; else export HENS_LIB=$GRAFT_REPO_ROOT/ab_live/libhens_$L.so; fi
  echo -n "$L: "; timeout 200 python tools/quick_bench.py --T $T --W $W --D $D --steps 400 --prof 1 2>&1 | sed -n 3,3p
done
