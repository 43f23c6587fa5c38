#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06
mkdir -p $out
./tools/batch1_kernels.sh $out 4 8 16 32 64 > /dev/null 2>&1
for w in 4 8 16 32 64; do echo "== width $w"; cat $out/b1_w${w}_timeline.txt; done
python tools/batch1_latency.py 2>/dev/null | grep -v "f32_small = 0" > $out/batch1_latency.txt
cat $out/batch1_latency.txt
