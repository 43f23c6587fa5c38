#!/bin/bash
# Collects the per-round profile evidence on the GPU box into gpurun_out/profiles_<round>/ (copy what is to be judged into profiles/):
#   rocprofv3 --kernel-trace --stats of the bench command per width and arithmetic (the exact-f32 reference arithmetic first),
#   separate --pmc passes (counters only) for MFMA utilisation, LDS conflicts and HBM traffic of FC 8x8 and conv 16x16 on both
#   arithmetics, the step timelines, the f32 tile sweep, and the default bench line with its detail file.
#   usage: tools/profile_round.sh r04        (from the repo root, ~6 minutes)
export TMPDIR=/tmp
r=${1:-r04}
out=gpurun_out/profiles_$r
mkdir -p $out
B="--steps 20 --warmup 3 --no-cpu-baseline --no-extras"
for wl in fc8 conv16 fc4 conv32 conv64; do
  for ar in f32 split; do
    # kernel durations with the branches of a conv pass on ONE stream (two kernels side by side share the chip and each one's
    # begin -> end says little); bench.py's own per-launch timing (roofline.achieved) runs on one stream too
    PNN_BRANCH_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 bench.py --workload $wl --arithmetic $ar $B > $out/${wl}_${ar}_trace.log 2>&1
    python3 tools/trace_summary.py $out/t > $out/${wl}_${ar}_kernel_summary.txt
    cp $out/t/*/*_kernel_stats.csv $out/${wl}_${ar}_kernel_stats.csv 2>/dev/null
    rm -rf $out/t
  done
done
for wl in fc8 conv16; do
  for ar in f32 split; do
    i=0
    for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
               "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
               "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      PNN_AUTOTUNE=0 PNN_BRANCH_STREAMS=0 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${wl}_${ar}/p$i -- python3 bench.py --workload $wl --arithmetic $ar --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $out/${wl}_${ar}_pmc_p$i.log 2>&1
    done
    python3 tools/pmc_summary.py ${wl}_${ar} > $out/${wl}_${ar}_pmc_summary.txt
    # timeline of one steady-state step (rule-based tiles: no tuning launches in the trace), default stream layout
    PNN_AUTOTUNE=0 rocprofv3 --kernel-trace --output-format csv -d $out/tl -- python3 bench.py --workload $wl --arithmetic $ar $B > /dev/null 2>&1
    python3 tools/trace_gaps.py $out/tl > $out/${wl}_${ar}_step_timeline.txt 2>&1
    rm -rf $out/tl
  done
done
python3 tools/pmc_traffic.py $out/pmc_traffic.json fc8_split conv16_split fc8_f32 conv16_f32 > /dev/null
python3 tools/f32_sweep.py fc8 conv16 fc4 conv32 conv64 > $out/f32_tile_sweep.txt 2>&1
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
cp bench_detail.json $out/bench_default_detail.json
rm -f $out/*_trace.log $out/*_pmc_p*.log
ls -la $out
