#!/bin/bash
# tails + captured launch chains (option graphs): the stress and the HM tests
mkdir -p gpurun_out/r06
PNN_GRAPHS=1 timeout 300 python3 tools/tails_stress.py 30 1 2>&1 | grep "tails ="
PNN_GRAPHS=1 timeout 1200 python3 -m pytest tests/test_hm.py tests/test_gpu_parity.py -m gpu -q -x -k "hm or graphs or tails or small" 2>&1 | tail -4
