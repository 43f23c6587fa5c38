#!/bin/bash
# round 6, experiment 2: K-segmented FC hidden layers (4 chains side by side at small M, seg_seq at batch) against round 5's library on one box
export TMPDIR=/tmp
out=gpurun_out/r06
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f32_small_kernel_bit_identical or one_summation_order or random_batch_sizes or full_size or fc_matches" > $out/exp2_tests.txt 2>&1
tail -5 $out/exp2_tests.txt
P=$PWD/tools/_bin/libpnn_hip_prev.so
D=$PWD/tools/_bin/libpnn_hip_diag.so
for n in 1 6; do
for i in 1 2; do
python tools/b1_opts.py --widths 4,8 --n $n --rounds 3 - 2>&1 | grep "^width" | sed 's/^/new  /'
PNN_LIB_PATH=$P python tools/b1_opts.py --widths 4,8 --n $n --rounds 3 - 2>&1 | grep "^width" | sed 's/^/prev /'
done
done > $out/exp2_b1.txt 2>&1
cat $out/exp2_b1.txt
for w in 4 8; do
PNN_LIB_PATH=$D PNN_B1_STAMPS=200 python tools/b1_opts.py --widths $w --rounds 1 - 2>&1 | grep "pnn-stamps\|^width"
done > $out/exp2_stamps.txt 2>&1
cat $out/exp2_stamps.txt
./tools/ab.sh run fc8 fc4 > $out/exp2_ab_bench.txt 2>&1
cat $out/exp2_ab_bench.txt
