#!/bin/bash
# XCD-affine numbering of the first launch's workgroups (default) against the plain numbering (HENS_NO_XCD=1), same box:
# throughput and per-launch event times on three shapes.
export PYTHONPATH=.
for shape in "16 4096 32" "16 16384 32" "8 16384 64"; do
  set -- $shape
  for x in 0 1 0 1; do
    if [ $x = 0 ]; then export HENS_NO_XCD=1; else unset HENS_NO_XCD; fi
    echo -n "shape $shape xcd-affine $x: "
    python tools/quick_bench.py --T $1 --W $2 --D $3 --steps 4000 --prof 1 | sed -n '1p;3p' | tr '\n' ' '; echo
  done
done
