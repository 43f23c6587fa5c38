#!/bin/bash
# Collects the per-round profile evidence on the GPU box into gpurun_out/profiles_<round>/ (copy what is to be judged into profiles/):
#   rocprofv3 --kernel-trace --stats of the bench command per width and arithmetic (the exact-f32 reference arithmetic first),
#   separate --pmc passes (counters only) for MFMA utilisation, LDS conflicts and HBM traffic of FC 8x8 and conv 16x16 on both
#   arithmetics, the step timelines, the f32 tile sweep, and the default bench line with its detail file.
#   usage: tools/profile_round.sh r06        (from the repo root, ~8 minutes)
export TMPDIR=/tmp
r=${1:-r06}
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
# round 5: the single-block path on the reference's arithmetic, the host-array entry points, the width workers side by side
( echo "# tools/batch1_latency.py, one MI355X (host call pnn_predict_pel at batch 1: staging + net + epilogue + wait); the library's defaults (plain launches)"; PNN_GRAPHS=0 python3 tools/batch1_latency.py 2>&1 | grep -v amdgpu.ids
  echo "# the same with PNN_GRAPHS=1 (option graphs: the launch chain of a shape captured once, replayed with one hipGraphLaunch)"; PNN_GRAPHS=1 python3 tools/batch1_latency.py 2>&1 | grep -v amdgpu.ids ) > $out/batch1_latency.txt
( echo "# tools/host_rate.py on one MI355X: the batched HOST-array entry points (pnn_predict_pel: host arrays in, int32 blocks out, one synchronous call per batch)"; python3 tools/host_rate.py fc8 conv16 fc4 conv32 conv64 2>&1 | grep -v amdgpu.ids
  echo "# ... and ONE call of 16 bench batches (--slices 16): slices overlapped (host_slice 0) against one copy in / passes / one copy out (-1)"; python3 tools/host_rate.py --slices 16 fc8 conv16 2>&1 | grep -v amdgpu.ids ) > $out/host_rate.txt
( echo "# tools/corun_threads.cpp: the batching service's five width workers as five host threads with one context each, configs[3]'s mean batches"; python3 tools/corun_threads.py 1.5 2>&1 | grep -v amdgpu.ids; python3 tools/corun_threads.py 1.5 queues 2>&1 | grep -v amdgpu.ids ) > $out/corun_widths.txt
[ -x tools/_bin/hwq_probe ] && ( timeout 60 ./tools/_bin/hwq_probe 10 ) > $out/hwq_probe.txt 2>&1
[ -x tools/_bin/f32_chain_probe ] && ( echo "# tools/f32_chain_probe.hip: one wave, one dependent accumulation chain per instruction form (cycles by s_memtime)"; ./tools/_bin/f32_chain_probe ) > $out/f32_chain_probe.txt
[ -x tools/_bin/corun_noise ] && python3 - > $out/corun_noise.txt 2>/dev/null <<'PY'
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools", "hm"))
import run_hm
with tempfile.TemporaryDirectory() as d:
    table, _ = run_hm.make_models(os.path.join(d, "models"))
    print("# tools/corun_noise.hip (one MI355X box; FC 4x4 model of the campaigns, seeded random init)")
    print(subprocess.run(["timeout", "120", "./tools/_bin/corun_noise", table, "1.0"], capture_output=True, text=True).stdout)
PY
[ -x tools/_bin/aql_probe ] && ( echo "# tools/aql_probe.cpp (one MI355X box)"; timeout 90 ./tools/_bin/aql_probe tools/_bin/aql_probe.hsaco 0.5 ) > $out/aql_probe.txt 2>&1
PNN_PRECISION=0 PNN_GRAPHS=0 ./tools/batch1_kernels.sh $out/b1_f32 4 8 16 32 64 > /dev/null 2>&1
for w in 4 8 16 32 64; do cp $out/b1_f32/b1_w${w}_timeline.txt $out/batch1_w${w}_f32_timeline.txt 2>/dev/null; done
rm -rf $out/b1_f32
# round 6: device-side stamps of one single-block call per width (diagnostic library, if it was shipped in tools/_bin), the per-layer table of the conv nets at batch
[ -f tools/_bin/libpnn_hip_diag.so ] && for w in 4 8 16 32 64; do PNN_LIB_PATH=$PWD/tools/_bin/libpnn_hip_diag.so PNN_B1_STAMPS=200 python3 tools/b1_opts.py --widths $w --rounds 1 - 2>&1 | grep "pnn-stamps"; done > $out/b1_stamps.txt
python3 tools/conv_layers.py conv16 conv32 > $out/conv_f32_layers.txt 2>&1
( for n in 1 6; do python3 tools/b1_opts.py --widths 4,8,16,32,64 --n $n --rounds 3 --calls 150 - chain_io=0 2>&1 | grep "^width"; done ) > $out/b1_chain_io.txt
rm -f $out/*_trace.log $out/*_pmc_p*.log
ls -la $out
