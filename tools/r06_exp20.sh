#!/bin/bash
mkdir -p gpurun_out/r06
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tails or five_contexts" 2>&1 | tail -4
timeout 300 python3 tools/tails_stress.py 40 1 2>&1 | grep "tails ="
for n in 1 6 20; do timeout 300 python3 tools/b1_opts.py --conv-small --widths 4 --n $n --rounds 3 --calls 200 tails=0 tails=1 2>&1 | grep "width\|rror"; done
