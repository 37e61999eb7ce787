#!/bin/bash
# Build ab_live/libhens_<name>.so (default: base) from the sources of a git revision (default HEAD) for a same-box A/B:
#   tools/mkbase.sh [rev] [name]      run with HENS_LIB=$GRAFT_REPO_ROOT/ab_live/libhens_<name>.so
REV=${1:-HEAD}; NAME=${2:-base}
R=$(git rev-parse --show-toplevel)
T=$(mktemp -d)
git -C $R worktree add -f --detach $T $REV > /dev/null 2>&1 || { echo "worktree failed"; exit 1; }
mkdir -p $R/ab_live
(cd $T && HENS_LIB=$R/ab_live/libhens_$NAME.so python -m eryn_amd._build > /dev/null) && echo "built ab_live/libhens_$NAME.so from $REV"
git -C $R worktree remove --force $T
