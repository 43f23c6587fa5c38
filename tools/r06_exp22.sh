#!/bin/bash
# few clients: both sides of a request poll before they sleep (ShmDoors::spin_us, 200 us while at most 20 slots are listed) -- round trips, then the
# natural campaigns with their host-cores leg, polling off (PNN_SERVICE_SPIN_SLOTS=0) and on
mkdir -p gpurun_out/r06
for slots in 0 20; do
  for spec in "4 4000 1" "8 4000 1" "8 4000 2" "16 3000 1" "32 2000 1"; do
    echo "PNN_SERVICE_SPIN_SLOTS=$slots service_rtt $spec: $(PNN_SERVICE_SPIN_SLOTS=$slots timeout 300 python3 tools/service_rtt.py $spec 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
  done
done > gpurun_out/r06/exp22b_rtt.txt 2>&1
cat gpurun_out/r06/exp22b_rtt.txt
for slots in 0 20; do
  for cfg in kodak bsds; do
    PNN_SERVICE_SPIN_SLOTS=$slots python3 bench.py --workload hm_$cfg --hm-pictures natural --arithmetic f32 --detail-file /tmp/d.json 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['hm']; c=d.get('cpu_baseline') or {}
print('spin_slots=$slots $cfg natural: wall %.2f s, decode==enc %s, service cpu %s s, host-cores leg: %s' % (h['wall_s_all_encodes_and_decodes'], h['every_decode_equals_its_encoder'], (h.get('host_cpu') or {}).get('service_cpu_s'), json.dumps(c)[:300]))"
  done
done > gpurun_out/r06/exp22b_campaigns.txt 2>&1
cat gpurun_out/r06/exp22b_campaigns.txt
