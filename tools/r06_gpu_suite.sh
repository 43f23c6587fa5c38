#!/bin/bash
# the whole -m gpu suite + the default bench line on one box (what the driver runs at round end)
export TMPDIR=/tmp
out=gpurun_out/r06
mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1
tail -5 $out/gpu_tests.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 4200 $out/bench_default.json
cp bench_detail.json $out/bench_default_detail.json 2>/dev/null
