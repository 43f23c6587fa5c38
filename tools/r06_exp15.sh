#!/bin/bash
mkdir -p gpurun_out/r06
SLICE_TRACE_CALLS=150 python3 tools/slice_trace.py run conv16 2>&1 | grep call > gpurun_out/r06/exp15_conv16_calls.txt
SLICE_TRACE_CALLS=300 python3 tools/slice_trace.py run fc8 2>&1 | grep call > gpurun_out/r06/exp15_fc8_calls.txt
for f in gpurun_out/r06/exp15_conv16_calls.txt gpurun_out/r06/exp15_fc8_calls.txt; do echo $f; awk '{print $2}' $f | tr '\n' ' ' | fold -w 200; echo; done
