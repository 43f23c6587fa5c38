#!/bin/bash
# round 6, experiment 8: host-array calls of 16 bench batches, slices overlapped against one copy in / passes / one copy out
export TMPDIR=/tmp
out=gpurun_out/r06
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "several_slices or predict_by_batch or f16_range" > $out/exp8_tests.txt 2>&1
tail -3 $out/exp8_tests.txt
python tools/host_rate.py --slices 16 fc8 conv16 fc4 > $out/exp8_host_rate_slices.txt 2>&1
cat $out/exp8_host_rate_slices.txt
python tools/host_rate.py fc8 conv16 > $out/exp8_host_rate.txt 2>&1
cat $out/exp8_host_rate.txt
