#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
build_ab/permlane_probe
export HENS_LIB=$PWD/build_ab/libhens_mf.so
timeout 600 python -m pytest -x -q -m gpu "tests/test_hip_parity.py::test_seeded_teacher_forced" "tests/test_hip_replay.py::test_replay_config2_full_size" -k "32 or config2" > $O/tests.log 2>&1
timeout 120 python tools/short_call.py > $O/short.log 2>&1
timeout 120 python tools/trace_fused.py 16 4096 32 3 1 2>&1 | grep "phase durations\|lifetime mean" > $O/trace.txt
timeout 120 python tools/trace_fused.py 16 4096 32 1 1 2>&1 | grep "phase durations\|lifetime mean" >> $O/trace.txt
export HENS_LIB=$PWD/build_ab/libhens_aql.so
timeout 120 python tools/short_call.py > $O/short_base.log 2>&1
timeout 120 python tools/trace_fused.py 16 4096 32 3 1 2>&1 | grep "phase durations\|lifetime mean" > $O/trace_base.txt
timeout 120 python tools/trace_fused.py 16 4096 32 1 1 2>&1 | grep "phase durations\|lifetime mean" >> $O/trace_base.txt
tail -n 12 $O/tests.log; cat $O/short.log $O/trace.txt; echo BASE; cat $O/short_base.log $O/trace_base.txt
