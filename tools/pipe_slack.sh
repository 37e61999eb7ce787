#!/bin/bash
# Slack of a pipeline rank's first launch on the reference's adaptation schedule: the previous sweep's swap-count flags are made to
# count as raised only n shader cycles after the adapting workgroup's start (HENS_PIPE_INJECT_CYCLES, dev builds: tools/devbuild.sh
# inj64 -DHENS_DEV_D=64; tools/devbuild.sh inj32) - iteration time against injected delay.  Beyond the slack every further cycle of
# delay is a cycle of iteration time, so the slope of the tail also calibrates the shader clock.
#   tools/pipe_slack.sh T W D lib [iters]
export PYTHONPATH=$GRAFT_REPO_ROOT PIPE_DELAY=0
T=$1; W=$2; D=$3; L=$4; N=${5:-200}
export HENS_LIB=$GRAFT_REPO_ROOT/ab_live/libhens_$L.so
for c in 0 1 2000 4000 6000 8000 10000 12000 16000 20000 24000 32000 40000 60000; do
  echo -n "inject $c cycles: "; HENS_PIPE_INJECT_CYCLES=$c timeout 200 python tools/pipe_prof.py $T $W $D $N 2>&1 | grep "^pipe" | sed 's/^pipe *//; s/, cascade.*//'
done
