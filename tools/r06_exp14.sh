#!/bin/bash
# timeline of a sliced host call (conv16, fc8): where do the 10-17 % over 16 passes go?
mkdir -p gpurun_out/r06; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for wl in conv16 fc8; do
  rm -rf /tmp/sl_$wl
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/sl_$wl -- python3 tools/slice_trace.py run $wl > gpurun_out/r06/exp14_$wl.txt 2>&1
  n=13; [ $wl = fc8 ] && n=4
  python3 tools/slice_trace.py show /tmp/sl_$wl $n >> gpurun_out/r06/exp14_$wl.txt 2>&1
  ls /tmp/sl_$wl/*/ >> gpurun_out/r06/exp14_$wl.txt
done
cat gpurun_out/r06/exp14_conv16.txt gpurun_out/r06/exp14_fc8.txt
