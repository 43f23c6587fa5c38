#!/bin/bash
# round 6, experiment 10: mid-depth conv layers (1152 <= K < 2304) in min(taps, 4) K segments, against the commit before (prev), one box
export TMPDIR=/tmp
out=gpurun_out/r06
mkdir -p $out
B=$PWD/tools/_bin
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_summation_order or random_batch_sizes or f32_small_kernel_bit or f32_tiles_and_position or conv_matches or full_size" > $out/exp10_tests.txt 2>&1
tail -3 $out/exp10_tests.txt
for i in 1 2; do
for lib in prev new; do
  L=$B/libpnn_hip_prev.so; [ $lib = new ] && L=$PWD/context_adaptive_neural_network_based_prediction_amd/libpnn_hip.so
  for wl in conv16 conv32 conv64; do
    v=$(PNN_LIB_PATH=$L python3 bench.py --workload $wl --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g blocks/s  %.4f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
    echo "$lib $wl: $v"
  done
done; done > $out/exp10_ab.txt 2>&1
cat $out/exp10_ab.txt
for lib in prev new; do
  L=$B/libpnn_hip_prev.so; [ $lib = new ] && L=$PWD/context_adaptive_neural_network_based_prediction_amd/libpnn_hip.so
  for n in 1 6; do
  PNN_LIB_PATH=$L python tools/b1_opts.py --widths 16,32,64 --n $n --rounds 3 --calls 150 - 2>&1 | grep "^width" | sed "s/^/$lib /"
done; done > $out/exp10_b1.txt 2>&1
cat $out/exp10_b1.txt
