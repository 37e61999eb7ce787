#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
python tools/trace_fused.py 16 4096 32 3 1 > $O/trace_f.txt 2>&1
python tools/trace_fused.py 16 4096 32 1 1 > $O/trace_s.txt 2>&1
bash tools/profile_bench.sh r4h_prof > $O/prof.log 2>&1
cat gpurun_out/r4h_prof/kernel_summary.txt | head; cat $O/trace_f.txt | head -40; cat $O/trace_s.txt | head -40
