#!/bin/bash
# round 6, experiment 6: the shared-memory request path of the batching service against the socket protocol, whole campaigns on one box
export TMPDIR=/tmp
out=gpurun_out/r06
mkdir -p $out
python -m pytest tests/test_hm.py -x -q -m gpu -k "batching_service or refuses or dealt" > $out/exp6_tests.txt 2>&1
tail -3 $out/exp6_tests.txt
for i in 1 2; do
for cfg in kodak bsds; do
for shm in 0 1; do
  PNN_SERVICE_SHM=$shm python3 tools/hm/campaign.py --config $cfg --picture-set synthetic > $out/exp6_${cfg}_shm${shm}_$i.json 2> $out/exp6_${cfg}_shm${shm}_$i.err
  python3 - $out/exp6_${cfg}_shm${shm}_$i.json $cfg $shm <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pw = d["service"]["per_width"]
print("%s shm=%s wall %.2f s x regular %s service cpu %s s all cpu %s s throttled %s mean batch %s | us/call %s | in-server us/request %s | queued %s" % (
    sys.argv[2], sys.argv[3], d["wall_s_all_encodes_and_decodes"], d.get("wall_vs_regular"), d["host_cpu"]["service_cpu_s"], d["host_cpu"].get("all_processes_cpu_s"), d["host_cpu"].get("times_throttled"),
    d["service"]["mean_batch"], [pw[k]["us_per_call"] for k in sorted(pw, key=int)], [pw[k].get("in_server_us_per_request") for k in sorted(pw, key=int)], [pw[k].get("queued_us_per_request") for k in sorted(pw, key=int)]), flush=True)
PY
done; done; done 2>&1 | tee $out/exp6_summary.txt
