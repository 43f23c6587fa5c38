#!/usr/bin/env python3
"""Timeline of the timed steps in a rocprofv3 --kernel-trace CSV directory: per step (from the launch behind a pass's last
kernel -- the reduction of an FC net, the last transposed convolution of a conv net -- to the next such launch; a conv pass
whose gather is fused into the image kernel has no gather launch to go by, and one whose last layer is fused in has no
last-layer launch: see below), the kernels in launch order with their
duration and the idle gap in front of each."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("pnn::", "").replace("void ", "")[:40]))
rows.sort()
LAST = ("fuse_reduce", "tconv_cout1")
starts = [i for i, r in enumerate(rows) if i > 0 and rows[i - 1][2].startswith(LAST) and not r[2].startswith(LAST)]
if len(starts) < 3:
    # a conv pass whose last layer runs inside the image kernel in front of it (fuse_tail) has no last-layer launch either:
    # behind the merger come the transposed-convolution stack, which ends in an image-kernel launch, and then the next pass,
    # which begins with one -- a step starts at the SECOND image-kernel launch after a merger
    starts = []
    seen = -1
    for i, r in enumerate(rows):
        if r[2].startswith("merger"):
            seen = 0
        elif seen >= 0 and r[2].startswith("convimg_sp_kernel"):
            seen += 1
            if seen == 2:
                starts.append(i)
                seen = -1
# the timed steps are the longest run of equally long gather-to-gather segments: take the 6 segments before the last 2
segs = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
# the timed steps are the bulk of the equally long segments (0.4 s of ramp-up steps + the timed regions); the per-launch-timed
# passes of the roofline section at the end have the same kernels with event gaps between them: take the 6 segments around
# the MEDIAN span, not the last ones
cand = [s for s in segs if s[1] - s[0] == min(b - a for a, b in segs)]
cand.sort(key=lambda s: rows[s[1]][0] - rows[s[0]][0])
mid = len(cand) // 2
pick = sorted(cand[max(0, mid - 3):mid + 3])
tot_busy = tot_span = tot_union = 0
for a, b in pick:
    tot_span += rows[b][0] - rows[a][0]
    tot_busy += sum(r[1] - r[0] for r in rows[a:b])
    end = rows[a][0]                                 # time with at least one kernel running (kernels of two streams overlap)
    for r in rows[a:b]:
        tot_union += max(0, r[1] - max(end, r[0]))
        end = max(end, r[1])
a, b = pick[len(pick) // 2]                      # the step that is printed: one from the middle
t0 = rows[a][0]
prev_end = rows[a - 1][1]
overlap = tot_busy > 1.02 * tot_union           # a few ns of timestamp overlap between back-to-back kernels are not concurrency
print("one step (of %d averaged):" % len(pick))
for r in rows[a:b]:
    if overlap:                                      # start relative to the step's first kernel: concurrent kernels show as such
        print("  start %6.1f us | %-40s %7.1f us" % ((r[0] - t0) / 1e3, r[2], (r[1] - r[0]) / 1e3))
    else:
        print("  gap %6.1f us | %-40s %7.1f us" % ((r[0] - prev_end) / 1e3, r[2], (r[1] - r[0]) / 1e3))
    prev_end = max(prev_end, r[1])
n = len(pick)
if overlap:
    print("mean step span %.1f us, sum of kernel durations %.1f us (kernels of two streams run side by side), no kernel running %.1f us"
          % (tot_span / n / 1e3, tot_busy / n / 1e3, (tot_span - tot_union) / n / 1e3))
else:
    print("mean step span %.1f us, kernels busy %.1f us, idle between kernels %.1f us" % (tot_span / n / 1e3, tot_busy / n / 1e3, (tot_span - tot_busy) / n / 1e3))
