#!/bin/bash
# round 6, experiment 3: the fold of the FC K segments at batch -- as a block between stages (A) or inside the next stage's first MFMAs (B) -- against round 5 (prev)
export TMPDIR=/tmp
out=gpurun_out/r06
mkdir -p $out
B=$PWD/tools/_bin
PNN_LIB_PATH=$B/libpnn_hip_foldB.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_summation_order or random_batch_sizes or full_size" > $out/exp3_tests.txt 2>&1
tail -3 $out/exp3_tests.txt
for i in 1 2 3; do
  for lib in prev foldA foldB; do
    for wl in fc8 fc4; do
      v=$(PNN_LIB_PATH=$B/libpnn_hip_$lib.so python3 bench.py --workload $wl --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g blocks/s  %.4f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
      echo "$lib $wl: $v"
    done
  done
done > $out/exp3_ab.txt 2>&1
cat $out/exp3_ab.txt
