#!/bin/bash
# round 6, experiment 11: chain-order activations between the small exact-f32 kernels (one 16-byte LDS-DMA per chunk instead of four 4-byte ones)
export TMPDIR=/tmp
out=gpurun_out/r06
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f32_small_kernel_bit or one_summation_order or random_batch_sizes or conv_matches or fc_matches or graphs" > $out/exp11_tests.txt 2>&1
tail -3 $out/exp11_tests.txt
for i in 1 2; do
for n in 1 6; do
python tools/b1_opts.py --widths 4,8,16,32,64 --n $n --rounds 3 --calls 150 - chain_io=0 2>&1 | grep "^width"
done; done > $out/exp11_b1.txt 2>&1
cat $out/exp11_b1.txt
