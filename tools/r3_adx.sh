#!/bin/bash
# the count rows' single reader (wave ADX) against both adaptation waves reading them (base = HEAD~), and the number of count
# rows (HENS_ACC_GROUPS x 8) under the single reader
export PYTHONPATH=.
for rep in 1 2; do
for L in base adx; do
  for g in 0 1 2; do
  echo -n "$L groups=$g: "
  HENS_ACC_GROUPS=$g HENS_LIB=build_ab/libhens_$L.so python tools/quick_bench.py --T 16 --W 4096 --D 32 --steps 4000 --prof 1 | sed -n '1p;2p' | tr '\n' ' ' | sed 's/T=16 W=4096 D=32: 4000 iters in//' | sed "s/'pt_ms.*n_iters': 4000,//"; echo
  done
done; done
