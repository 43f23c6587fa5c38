#!/bin/bash
# the 4x4 FC net's output layer as the tail of the last hidden layer: bits, stress, then per-call time against tails=0
mkdir -p gpurun_out/r06
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tail or five_contexts" 2>&1 | tail -5
timeout 300 python3 tools/tails_stress.py 40 1 2>&1 | grep "tails ="
for n in 1 6 20; do timeout 300 python3 tools/b1_opts.py --widths 4 --n $n --rounds 5 --calls 300 tails=0 tails=1 2>&1 | grep "width\|rror"; done
