#!/bin/bash
mkdir -p gpurun_out/r06
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tails" 2>&1 | tail -3
for rep in 1 2; do
  for t in 1 0; do timeout 200 python3 tools/tails_stress.py 25 $t 2>&1 | grep "tails ="; done
done > gpurun_out/r06/exp18_stress.txt 2>&1
cat gpurun_out/r06/exp18_stress.txt
