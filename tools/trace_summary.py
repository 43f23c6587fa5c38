#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV directory: per (kernel, grid) call count and mean duration."""
import collections, csv, glob, sys
d = sys.argv[1]
agg = collections.OrderedDict()
tot = 0.0
for f in glob.glob(d + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].replace("pnn::", "").replace("(TapGemmParams)", "").replace("void ", "")[:44],
             r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg.setdefault(k, []).append(dt)
        tot += dt
print("%-44s %9s %5s %3s %6s %10s %8s" % ("kernel", "grid_x", "y", "z", "calls", "avg_us", "share"))
for k, v in agg.items():
    print("%-44s %9s %5s %3s %6d %10.1f %7.1f%%" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), 100 * sum(v) / tot))
