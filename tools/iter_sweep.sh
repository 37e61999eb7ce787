#!/bin/bash
# one-launch iteration (k_iter) vs two launches (HENS_NO_ITER=1) over shapes, same box
export PYTHONPATH=$GRAFT_REPO_ROOT
for shp in "4 4096 32" "8 4096 32" "16 2048 32" "32 1024 32" "16 4096 16" "8 8192 32" "16 4096 32" "32 2048 32" "16 6144 32"; do
  set -- $shp
  a=$(HENS_ITER_MAX=64 timeout 120 python tools/quick_bench.py --T $1 --W $2 --D $3 --steps 1500 2>&1 | grep -o '[0-9.]* us/iter')
  b=$(HENS_NO_ITER=1 timeout 120 python tools/quick_bench.py --T $1 --W $2 --D $3 --steps 1500 2>&1 | grep -o '[0-9.]* us/iter')
  echo "T=$1 W=$2 D=$3: k_iter $a   two launches $b"
done
