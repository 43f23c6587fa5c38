#!/bin/bash
# Collects the per-round profile evidence on the GPU box into gpurun_out/profiles_<round>/ :
#   rocprofv3 --kernel-trace --stats of the default bench command (FC 8x8) and of conv16, plus PMC passes
#   (separate runs, --pmc only) for MFMA utilisation, LDS conflicts and HBM traffic, and the step timeline.
export TMPDIR=/tmp
r=${1:-r01}
out=gpurun_out/profiles_$r
mkdir -p $out
# kernel durations are taken with the branches of a conv pass on ONE stream (PNN_BRANCH_STREAMS=0): side by side on two
# streams (the default at batch) two kernels share the chip and each one's begin -> end says little;
# bench.py's own per-launch timing (roofline.achieved) runs on one stream too.  conv16_step_timeline_overlap.txt is the default.
export PNN_BRANCH_STREAMS=0
for wl in fc8 conv16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${wl}_trace -- python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-sustained > $out/${wl}_trace.log 2>&1
  python3 tools/trace_summary.py $out/${wl}_trace > $out/${wl}_kernel_summary.txt
  cp $out/${wl}_trace/*/*_kernel_stats.csv $out/${wl}_kernel_stats.csv
  # the same on the exact-f32 kernels (the reference's arithmetic): the rocprof side of bench.py's `reference_arithmetic` roofline
  PNN_PRECISION=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${wl}_f32_trace -- python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-sustained > $out/${wl}_f32_trace.log 2>&1
  python3 tools/trace_summary.py $out/${wl}_f32_trace > $out/${wl}_f32_kernel_summary.txt
  cp $out/${wl}_f32_trace/*/*_kernel_stats.csv $out/${wl}_f32_kernel_stats.csv
  rm -rf $out/${wl}_f32_trace
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    PNN_AUTOTUNE=0 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${wl}/p$i -- python3 bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-sustained > $out/${wl}_pmc_p$i.log 2>&1
  done
  python3 tools/pmc_summary.py $wl > $out/${wl}_pmc_summary.txt
  rm -rf $out/${wl}_trace
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do          # HBM traffic of the exact-f32 kernels, its own passes
    i=$((i+1))
    PNN_PRECISION=0 PNN_AUTOTUNE=0 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${wl}_f32/p$((i+2)) -- python3 bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-sustained > $out/${wl}_f32_pmc_p$i.log 2>&1
  done
done
python3 tools/pmc_traffic.py $out/pmc_traffic.json fc8 conv16 fc8_f32 conv16_f32 > /dev/null
for wl in fc8 conv16; do   # timeline of one steady-state step (rule-based tiles: no tuning launches in the trace)
  PNN_AUTOTUNE=0 rocprofv3 --kernel-trace --output-format csv -d $out/${wl}_tl -- python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-sustained > /dev/null 2>&1
  python3 tools/trace_gaps.py $out/${wl}_tl > $out/${wl}_step_timeline.txt
  rm -rf $out/${wl}_tl
done
unset PNN_BRANCH_STREAMS
PNN_AUTOTUNE=0 rocprofv3 --kernel-trace --output-format csv -d $out/conv16_tl2 -- python3 bench.py --workload conv16 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-sustained > /dev/null 2>&1
python3 tools/trace_gaps.py $out/conv16_tl2 > $out/conv16_step_timeline_overlap.txt
rm -rf $out/conv16_tl2
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
ls -la $out
