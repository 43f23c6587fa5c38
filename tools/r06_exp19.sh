#!/bin/bash
# after the tile-major fix: a long stress, the whole GPU suite, then configs[3] / configs[4] on natural pictures with tails off and on (same box)
mkdir -p gpurun_out/r06
timeout 400 python3 tools/tails_stress.py 120 1 2>&1 | grep "tails =" > gpurun_out/r06/exp19_stress.txt
cat gpurun_out/r06/exp19_stress.txt
./tools/r06_gpu_suite.sh > gpurun_out/r06/exp19_suite.txt 2>&1
grep -v "^{" gpurun_out/r06/exp19_suite.txt | tail -4
for rep in 1 2; do
for t in 0 1; do
  for cfg in kodak bsds; do
    PNN_TAILS=$t python3 bench.py --workload hm_$cfg --hm-pictures natural --arithmetic f32 --no-cpu-baseline --detail-file /tmp/d.json 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['hm']
print('tails=$t $cfg natural: wall %.2f s, decode==enc %s, service cpu %s s' % (h['wall_s_all_encodes_and_decodes'], h['every_decode_equals_its_encoder'], (h.get('host_cpu') or {}).get('service_cpu_s')))"
  done
done
done > gpurun_out/r06/exp19_campaigns.txt 2>&1
cat gpurun_out/r06/exp19_campaigns.txt
