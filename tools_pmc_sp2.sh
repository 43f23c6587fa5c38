#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/sp2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/sp2/p1 -- python3 tools/sp_prof.py 0 ${1:-4096} > gpurun_out/sp2_p1.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d gpurun_out/sp2/p2 -- python3 tools/sp_prof.py 0 ${1:-4096} > gpurun_out/sp2_p2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
pm=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/sp2/p*/*/*_counter_collection.csv'):
    per=collections.defaultdict(float); meta={}
    for r in csv.DictReader(open(f)):
        per[(r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value']); meta[r['Dispatch_Id']]=(r['Kernel_Name'][:44],r['Grid_Size'])
    for (d,cn),v in per.items(): pm[meta[d]][cn].append(v)
for k in pm:
    if 'tapgemm' in k[0]: print(k, {cn: ['%.3g'%x for x in v[-3:]] for cn,v in pm[k].items()})
PY
tail -3 gpurun_out/sp2_p2.log
