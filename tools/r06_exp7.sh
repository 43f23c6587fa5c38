#!/bin/bash
# round 6, experiment 7: FC hidden layers with all operands resident (fcseg_f32_small_all_kernel) + per-workgroup completion flags of fc_out, against the commit before (prev)
export TMPDIR=/tmp
out=gpurun_out/r06
mkdir -p $out
B=$PWD/tools/_bin
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f32_small_kernel_bit or one_summation_order or random_batch_sizes or graphs or fc_matches" > $out/exp7_tests.txt 2>&1
tail -3 $out/exp7_tests.txt
for i in 1 2; do
for n in 1 6 40; do
python tools/b1_opts.py --widths 4,8 --n $n --rounds 3 - 2>&1 | grep "^width" | sed 's/^/new      /'
PNN_FCSEG_RING=1 python tools/b1_opts.py --widths 4,8 --n $n --rounds 3 - 2>&1 | grep "^width" | sed 's/^/new+ring /'
PNN_LIB_PATH=$B/libpnn_hip_prev.so python tools/b1_opts.py --widths 4,8 --n $n --rounds 3 - 2>&1 | grep "^width" | sed 's/^/prev     /'
done; done > $out/exp7_b1.txt 2>&1
cat $out/exp7_b1.txt
for w in 4 8; do
PNN_LIB_PATH=$B/libpnn_hip_diag.so PNN_B1_STAMPS=200 python tools/b1_opts.py --widths $w --rounds 1 - 2>&1 | grep "pnn-stamps\|^width"
done > $out/exp7_stamps.txt 2>&1
cat $out/exp7_stamps.txt
