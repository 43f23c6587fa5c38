#!/usr/bin/env python3
"""Timeline of ONE single-block call from a rocprofv3 --kernel-trace of tools/batch1_trace.py: the kernels in launch order
with duration and the idle gap in front of each, plus the mean span / busy time per call.  usage: batch1_timeline.py <dir> <calls>"""
import csv, glob, sys
d, calls = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("pnn::", "").replace("void ", "")[:52],
                     r["Grid_Size_X"], r["Workgroup_Size_X"]))
rows.sort()
per = len(rows) // calls
rows = rows[len(rows) - per * (calls // 2):]            # the second half of the calls: steady state
n = len(rows) // per
span = busy = 0
for c in range(n):
    seg = rows[c * per:(c + 1) * per]
    span += seg[-1][1] - seg[0][0]
    busy += sum(r[1] - r[0] for r in seg)
seg = rows[(n - 1) * per:]
prev = None
print("one call = %d launches:" % per)
for r in seg:
    gap = 0.0 if prev is None else (r[0] - prev) / 1e3
    print("  gap %5.1f us | %-52s grid %6s x %4s %6.1f us" % (gap, r[2], r[3], r[4], (r[1] - r[0]) / 1e3))
    prev = r[1]
print("mean over %d calls: first kernel start -> last kernel end %.1f us, kernels busy %.1f us" % (n, span / n / 1e3, busy / n / 1e3))
