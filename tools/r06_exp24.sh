#!/bin/bash
# the FC 4x4 output tail inside campaigns: synthetic sets (FC nets for 4x4 / 8x8), conv tails only (PNN_TAILS=1) against all tails (3), alternated
mkdir -p gpurun_out/r06
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tail" 2>&1 | tail -2
for rep in 1 2 3; do
for t in 1 3; do
  for cfg in kodak bsds; do
    PNN_TAILS=$t python3 bench.py --workload hm_$cfg --hm-pictures synthetic --arithmetic f32 --no-cpu-baseline --detail-file /tmp/d.json 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['hm']
print('PNN_TAILS=$t $cfg synthetic: wall %.2f s, decode==enc %s, service cpu %s s' % (h['wall_s_all_encodes_and_decodes'], h['every_decode_equals_its_encoder'], (h.get('host_cpu') or {}).get('service_cpu_s')))"
  done
done
done > gpurun_out/r06/exp24_campaigns.txt 2>&1
cat gpurun_out/r06/exp24_campaigns.txt
