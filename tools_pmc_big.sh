#!/bin/bash
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/big/trace -- python3 bench.py --batch 262144 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/big_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/big/pmc -- python3 bench.py --batch 262144 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/big_pmc.log 2>&1
python3 - <<'PY'
import csv,glob,collections
tr=collections.defaultdict(list)
for f in glob.glob('gpurun_out/big/trace/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        tr[(r['Kernel_Name'][:60],r['Grid_Size_X'],r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in tr.items():
    if 'tapgemm' in k[0]: print(k, len(v), 'avg_us=%.1f'%(sum(v)/len(v)))
pm=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/big/pmc/*/*_counter_collection.csv'):
    per=collections.defaultdict(float); meta={}
    for r in csv.DictReader(open(f)):
        per[(r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value']); meta[r['Dispatch_Id']]=(r['Kernel_Name'][:60],r['Grid_Size'])
    for (d,cn),v in per.items(): pm[meta[d]][cn].append(v)
for k in pm:
    if 'tapgemm' in k[0]:
        print(k, {cn: '%.4g'%(sum(v)/len(v)) for cn,v in pm[k].items()})
PY
