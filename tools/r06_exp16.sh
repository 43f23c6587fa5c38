#!/bin/bash
# final library: the new sliced-call edge-case test, then the eight campaigns (lines with the arithmetic tags of both sides)
mkdir -p gpurun_out/r06
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sliced_host_call_edge_cases or several_slices" > gpurun_out/r06/exp16_tests.txt 2>&1
tail -3 gpurun_out/r06/exp16_tests.txt
./tools/r06_campaigns.sh
