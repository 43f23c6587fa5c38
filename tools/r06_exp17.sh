#!/bin/bash
# merger / last layer as tails of the small GEMM launches: bits, then per-call time against tails=0
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tails" > gpurun_out/r06/exp17_tests.txt 2>&1
tail -15 gpurun_out/r06/exp17_tests.txt
for n in 1 3 6; do
  timeout 300 python3 tools/b1_opts.py --widths 16,32 --n $n --rounds 3 --calls 200 tails=0 tails=1 2>&1 | grep "width\|Error\|error"
  timeout 300 python3 tools/b1_opts.py --conv-small --widths 4,8 --n $n --rounds 3 --calls 200 tails=0 tails=1 2>&1 | grep "width\|Error\|error"
done > gpurun_out/r06/exp17_b1.txt 2>&1
cat gpurun_out/r06/exp17_b1.txt
