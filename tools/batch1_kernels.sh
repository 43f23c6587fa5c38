#!/bin/bash
# per-kernel trace of single-block calls: ./tools/batch1_kernels.sh <out dir> [widths...]   (canonical order on)
export TMPDIR=/tmp
out=$1; shift
mkdir -p $out
for w in ${@:-8 16}; do
  rm -rf /tmp/b1_$w
  rocprofv3 --kernel-trace --output-format csv -d /tmp/b1_$w -- python3 tools/batch1_trace.py $w 1 60 > $out/b1_w$w.log 2>&1
  python3 tools/batch1_timeline.py /tmp/b1_$w 60 > $out/b1_w${w}_timeline.txt 2>&1
  python3 tools/trace_summary.py /tmp/b1_$w > $out/b1_w${w}_kernels.txt 2>&1
done
