#!/bin/bash
# round 6, experiment 4: FC K segments of 640 (two per 1200-deep layer) instead of 320 (four): the fold at batch, the chains at small M
export TMPDIR=/tmp
out=gpurun_out/r06
mkdir -p $out
B=$PWD/tools/_bin
PNN_LIB_PATH=$B/libpnn_hip_seg40.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_summation_order or random_batch_sizes or f32_small_kernel_bit" > $out/exp4_tests.txt 2>&1
tail -3 $out/exp4_tests.txt
for i in 1 2 3; do
  for lib in prev foldA seg40; do
    for wl in fc8 fc4; do
      v=$(PNN_LIB_PATH=$B/libpnn_hip_$lib.so python3 bench.py --workload $wl --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g blocks/s  %.4f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
      echo "$lib $wl: $v"
    done
  done
done > $out/exp4_ab.txt 2>&1
cat $out/exp4_ab.txt
for lib in prev foldA seg40; do
for n in 1 6; do
PNN_LIB_PATH=$B/libpnn_hip_$lib.so python tools/b1_opts.py --widths 4,8 --n $n --rounds 3 - 2>&1 | grep "^width" | sed "s/^/$lib /"
done; done > $out/exp4_b1.txt 2>&1
cat $out/exp4_b1.txt
