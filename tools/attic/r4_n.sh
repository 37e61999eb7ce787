#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1
tail -n 14 $O/suite.log
bash tools/r4_final_profiles.sh r04a > $O/final.log 2>&1
tail -n 30 $O/final.log
