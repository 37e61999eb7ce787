#!/bin/bash
# Build ab_live/libhens_base.so from the sources of a git revision (default HEAD) for a same-box A/B (tools/ab_lib.sh).
REV=${1:-HEAD}
R=$(git rev-parse --show-toplevel)
T=$(mktemp -d)
mkdir -p $T/eryn_amd/csrc $T/include
for f in hens.hip hens_kernels.h hens_rj.h hens_iter.h; do git show $REV:eryn_amd/csrc/$f > $T/eryn_amd/csrc/$f 2>/dev/null; done
git show $REV:include/hipensemble.h > $T/include/hipensemble.h
mkdir -p $R/ab_live
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $T/eryn_amd/csrc/hens.hip -o $R/ab_live/libhens_base.so 2>/dev/null && echo built libhens_base.so from $REV
rm -rf $T
