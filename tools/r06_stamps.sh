#!/bin/bash
# device-side stamps of single-block calls with the final (diagnostic) library: three samples per width, conv 4x4 / 8x8 too
mkdir -p gpurun_out/r06
for rep in 1 2 3; do
  for w in 4 8 16 32 64; do PNN_LIB_PATH=$PWD/tools/_bin/libpnn_hip_diag.so PNN_B1_STAMPS=$((150 + 37 * rep)) python3 tools/b1_opts.py --widths $w --rounds 1 - 2>&1 | grep "pnn-stamps"; done
  for w in 4 8; do PNN_LIB_PATH=$PWD/tools/_bin/libpnn_hip_diag.so PNN_B1_STAMPS=$((150 + 37 * rep)) python3 tools/b1_opts.py --conv-small --widths $w --rounds 1 - 2>&1 | grep "pnn-stamps" | sed 's/^\[pnn-stamps\] width/[pnn-stamps] CONV width/'; done
done > gpurun_out/r06/b1_stamps3.txt
grep "pnn-stamps\] \(CONV \)\?width" gpurun_out/r06/b1_stamps3.txt | cut -c1-120
