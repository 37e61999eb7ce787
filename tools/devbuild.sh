#!/bin/bash
# Quick A/B build: only the dense-Gaussian D = 32 kernels (config 2).  usage: tools/devbuild.sh <name> [extra hipcc flags]
#   -> ab_live/libhens_<name>.so ; run with HENS_LIB=$PWD/ab_live/libhens_<name>.so
R=$(git rev-parse --show-toplevel); N=${1:-dev}; shift
mkdir -p $R/ab_live
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-cuda-compat -DHENS_DEV_BUILD "$@" -I$R/include \
    $R/eryn_amd/csrc/hens.hip $R/eryn_amd/csrc/hens_k_dense.hip -o $R/ab_live/libhens_$N.so -L/opt/rocm/lib -lhsa-runtime64 2>&1 | grep -E "error|Error" ; ls -la $R/ab_live/libhens_$N.so
