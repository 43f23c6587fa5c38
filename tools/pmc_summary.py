#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs written by tools/pmc.sh: per kernel (name, grid), mean counter values per dispatch."""
import collections, csv, glob, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_%s/p*/*/*_counter_collection.csv" % tag):
    per = collections.defaultdict(float)
    meta = {}
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per[key] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = (r["Kernel_Name"].replace("pnn::", "").replace("(TapGemmParams)", "")[:40], r["Grid_Size"])
    for (d, cn), v in per.items():
        agg[meta[d]][cn].append(v)
for k in sorted(agg):
    print(k)
    for cn in sorted(agg[k]):
        v = agg[k][cn]
        print("   %-28s n=%3d mean=%.4g" % (cn, len(v), sum(v) / len(v)))
