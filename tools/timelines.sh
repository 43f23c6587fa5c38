export TMPDIR=/tmp
mkdir -p gpurun_out/tl
for wl in fc8 conv16; do for at in 0 1; do
  PNN_AUTOTUNE=$at rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_${wl}_$at -- python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 tools/trace_gaps.py /tmp/tl_${wl}_$at > gpurun_out/tl/${wl}_at$at.txt
  echo "== $wl autotune=$at"; cat gpurun_out/tl/${wl}_at$at.txt
done; done
