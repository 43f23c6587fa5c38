#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/sp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sp/p1 -- python3 tools/sp_prof.py 0 65536 > gpurun_out/sp_p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/sp/p2 -- python3 tools/sp_prof.py 0 65536 > gpurun_out/sp_p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sp/tr -- python3 tools/sp_prof.py 0 65536 > gpurun_out/sp_tr.log 2>&1
python3 - <<'PY'
import csv,glob,collections
pm=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/sp/p*/*/*_counter_collection.csv'):
    per=collections.defaultdict(float); meta={}
    for r in csv.DictReader(open(f)):
        per[(r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value']); meta[r['Dispatch_Id']]=(r['Kernel_Name'][:50],r['Grid_Size'])
    for (d,cn),v in per.items(): pm[meta[d]][cn].append(v)
for k in pm:
    if 'tapgemm_sp' in k[0]: print(k, {cn: '%.4g'%(sum(v)/len(v)) for cn,v in pm[k].items()})
tr=collections.defaultdict(list)
for f in glob.glob('gpurun_out/sp/tr/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        tr[(r['Kernel_Name'][:50],r['Grid_Size_X'],r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in tr.items():
    if 'tapgemm_sp' in k[0]: print(k, len(v), 'avg_us=%.1f'%(sum(v)/len(v)))
PY
