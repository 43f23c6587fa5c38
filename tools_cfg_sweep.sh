#!/bin/bash
# usage: tools_cfg_sweep.sh <workload> <ncfgs>  -> per (layer shape, cfg) median launch time from PNN_PROFILE
wl=$1; n=$2
for c in $(seq 0 $((n-1))); do
  PNN_TILE_CFG=$c PNN_PROFILE=1 python bench.py --workload $wl --steps 6 --warmup 1 --no-cpu-baseline 2>&1 | grep "^\[pnn-prof\]"
done | python -c "
import sys,collections,statistics
d=collections.defaultdict(list)
for l in sys.stdin:
    kv=dict(x.split('=') for x in l.split()[1:])
    d[(kv['M'],kv['K'],kv['N'],kv['ncls'],kv['cfg'],kv['mf']+':'+kv['rt'],kv['nt'],kv['kc'])].append(float(kv['us']))
shapes=collections.defaultdict(list)
for k,v in d.items(): shapes[k[:4]].append((statistics.median(v),k[4:]))
for sh,lst in shapes.items():
    lst.sort()
    print('M=%s K=%s N=%s ncls=%s'%sh, ' | '.join('%.1fus cfg%s{%s,%s,%s}'%((t,)+c) for t,c in lst[:6]))
"
