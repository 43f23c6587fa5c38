#!/bin/bash
# round 6, experiment 9: K segments of the deep conv layers in sequence inside the workgroups (f32_seg_mode = 1, now on the two-stage inner loops) against parallel segments + reduce
export TMPDIR=/tmp
out=gpurun_out/r06
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f32_tiles_and_position" > $out/exp9_tests.txt 2>&1
tail -3 $out/exp9_tests.txt
for i in 1 2; do
for wl in conv32 conv64; do
for mode in 0 1 -1; do
  v=$(PNN_F32_SEG_MODE=$mode python3 bench.py --workload $wl --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g blocks/s  %.4f ms  frac %.3f launches %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launches_timed']))")
  echo "$wl f32_seg_mode=$mode: $v"
done; done; done > $out/exp9_segmode.txt 2>&1
cat $out/exp9_segmode.txt
