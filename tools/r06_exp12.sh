#!/bin/bash
# cin1 layer on the f32 matrix instruction: same bits as the VALU form?  parity, then A/B at batch and per single block
mkdir -p gpurun_out/r06
python3 tools/lib_ab_bits.py tools/_bin/libpnn_hip_prev.so context_adaptive_neural_network_based_prediction_amd/libpnn_hip.so > gpurun_out/r06/exp12_bits.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv or chain or merger or golden or natural" > gpurun_out/r06/exp12_tests.txt 2>&1
./tools/ab.sh run conv16 conv32 conv64 > gpurun_out/r06/exp12_ab.txt 2>&1
for lib in tools/_bin/libpnn_hip_prev.so context_adaptive_neural_network_based_prediction_amd/libpnn_hip.so; do
  echo "== $lib"; PNN_LIB_PATH=$PWD/$lib python3 tools/b1_opts.py --widths 16,32,64 --rounds 3 - 2>&1 | grep width
done > gpurun_out/r06/exp12_b1.txt 2>&1
tail -12 gpurun_out/r06/exp12_bits.txt; tail -3 gpurun_out/r06/exp12_tests.txt; cat gpurun_out/r06/exp12_ab.txt gpurun_out/r06/exp12_b1.txt
