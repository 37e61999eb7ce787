#!/bin/bash
O=gpurun_out/r4o; mkdir -p $O
export HENS_LIB=$PWD/build_ab/libhens_walk.so
timeout 600 python -m pytest -x -q -m gpu "tests/test_hip_parity.py::test_seeded_teacher_forced" "tests/test_hip_replay.py::test_replay_config2_full_size" tests/test_hip_repeat.py -k "32 and not 5x100" > $O/tests.log 2>&1
tail -n 6 $O/tests.log
for L in aql walk; do
export HENS_LIB=$PWD/build_ab/libhens_$L.so
echo "== $L"
timeout 120 python tools/short_call.py 2>&1 | grep "long call\|eng.sync"
timeout 120 python tools/trace_fused.py 16 4096 32 3 1 2>&1 | grep "phase durations\|lifetime mean"
done > $O/ab.txt 2>&1
cat $O/ab.txt
HENS_LIB=$PWD/build_ab/libhens_walk.so timeout 300 python tools/aql_check.py 16 4096 32 3000 20
