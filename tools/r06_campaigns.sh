#!/bin/bash
export TMPDIR=/tmp
./tools/hm/record_campaigns.sh r06 f32 split 2>&1 | tee gpurun_out/r06/hm_runs_raw.txt
python3 tools/hm/summarize_campaigns.py gpurun_out/r06 > gpurun_out/r06/hm_runs.txt 2>&1
cat gpurun_out/r06/hm_runs.txt
