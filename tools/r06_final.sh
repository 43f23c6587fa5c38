#!/bin/bash
# the round's final evidence from ONE box: profile collection, then the eight campaigns
./tools/profile_round.sh r06 > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log
./tools/r06_campaigns.sh > gpurun_out/campaigns.log 2>&1
tail -40 gpurun_out/r06/hm_runs.txt
