#!/bin/bash
# per-kernel stats of one bench workload: ./tools/kernel_stats.sh <workload> [extra bench args]
export TMPDIR=/tmp
wl=${1:-conv16}; shift
mkdir -p gpurun_out/ks
rm -rf /tmp/ks
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 bench.py --workload $wl --no-cpu-baseline --no-extras --steps 20 --warmup 3 "$@" > gpurun_out/ks/$wl.log 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-70s n=%6s avg=%9.1f us  %5s%%" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
