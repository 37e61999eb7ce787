#!/bin/bash
# same-box timing of dev builds at config 2: tools/ab_dev.sh [-e "ENV=1 ..."] [-a "quick_bench args"] name1 name2 ...   (ab_live/libhens_<name>.so)
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
EXTRA=""; ARGS=""
while [ "$1" == "-e" ] || [ "$1" == "-a" ]; do
  if [ "$1" == "-e" ]; then EXTRA="$2"; else ARGS="$2"; fi
  shift 2
done
for rep in 1 2; do
  for n in "$@"; do
    echo -n "$n: "; env $EXTRA HENS_LIB=$GRAFT_REPO_ROOT/ab_live/libhens_$n.so timeout 120 python tools/quick_bench.py --steps 4000 $ARGS 2>&1 | grep -o "[0-9.]* us/iter" || echo failed
  done
done
