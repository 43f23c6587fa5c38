#!/bin/bash
# BASELINE configs[3] / configs[4] at their stated picture counts through bench.py, on both arithmetics and both picture sets:
#   tools/hm/record_campaigns.sh r05 [arithmetics...]        (GPU box)  ->  gpurun_out/<round>/hm_<config>_<pictures>_<arithmetic>.json (+ _detail)
round=${1:-r05}; shift
ariths=${@:-f32 split}
out=gpurun_out/$round
mkdir -p $out
for cfg in kodak bsds; do
  for pics in synthetic natural; do
    for ar in $ariths; do
      extra=""
      [ "$ar" != "f32" ] && extra="--no-cpu-baseline"      # the PNN-on-host-cores leg once per campaign is enough
      python3 bench.py --workload hm_$cfg --hm-pictures $pics --arithmetic $ar $extra --detail-file $out/hm_${cfg}_${pics}_${ar}_detail.json \
        > $out/hm_${cfg}_${pics}_${ar}.json 2> $out/hm_${cfg}_${pics}_${ar}.err
      python3 - $out/hm_${cfg}_${pics}_${ar}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d["hm"]
print("%-28s %-5s wall %6.2f s  x regular %5.2f  decode==enc %s  service cpu %s s  all cpu %s s  throttled %s" % (
    d["config"]["workload"][:11] + " " + d["config"]["picture_set"], h.get("arithmetic"), h["wall_s_all_encodes_and_decodes"], h["wall_vs_regular"] or 0,
    h["every_decode_equals_its_encoder"], (h.get("host_cpu") or {}).get("service_cpu_s"), (h.get("host_cpu") or {}).get("all_processes_cpu_s"),
    (h.get("host_cpu") or {}).get("times_throttled")))
PY
    done
  done
done
