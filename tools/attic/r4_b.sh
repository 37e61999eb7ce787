#!/bin/bash
O=gpurun_out/r4b; mkdir -p $O
python tools/call_timeline.py 20 > $O/timeline20.txt 2>&1
python tools/call_timeline.py 3 > $O/timeline3.txt 2>&1
python tools/short_call.py > $O/short.log 2>&1
python -m pytest tests/test_hip_repeat.py -x -q -m gpu --durations=10 > $O/repeat.log 2>&1
python -m pytest tests -x -q -m gpu --deselect tests/test_hip_repeat.py > $O/suite.log 2>&1
tail -5 $O/repeat.log $O/suite.log; cat $O/timeline20.txt $O/timeline3.txt $O/short.log
