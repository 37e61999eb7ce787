#!/bin/bash
# round 4, session a: the fixed cost of a short call (probe + the library as it stands)
O=gpurun_out/r4a; mkdir -p $O
build_ab/call_floor 40 8 > $O/call_floor_40x8.txt 2>&1
build_ab/call_floor 40 2 > $O/call_floor_40x2.txt 2>&1
python tools/short_call.py > $O/short.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench20.json 2> $O/bench20.err
python bench.py --no-cpu > $O/bench_long.json 2> $O/bench_long.err
cat $O/call_floor_40x8.txt $O/short.log
