#!/usr/bin/env python3
"""GPU timeline of the last training step in a rocprofv3 --kernel-trace CSV: wall span, busy time (union of the kernel
intervals), idle gaps between consecutive kernels (count, total, median, how many are > 5 us) -- is a step kernel-bound,
launch-gap-bound or host-bound?   python tools/gpu_timeline.py <kernel_trace.csv> <kernels per step, or 0 = split at the largest gaps>"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# steps are separated by the host-side synchronisation of the timing loop: the `nsteps - 1` largest gaps of the tail
gaps = sorted(((rows[i + 1][0] - max(r[1] for r in rows[max(0, i - 8):i + 1]), i) for i in range(len(rows) // 2, len(rows) - 1)), reverse=True)
cuts = sorted(i for _, i in gaps[:max(1, nsteps - 1)])
lo, hi = cuts[-2] + 1 if len(cuts) >= 2 else cuts[-1] + 1, cuts[-1] + 1
step = rows[lo:hi] if len(cuts) >= 2 else rows[cuts[-1] + 1:]
span = step[-1][1] - step[0][0]
busy, end = 0, step[0][0]
g = []
for s, e, _ in step:
    if s > end:
        g.append(s - end)
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    end = max(end, e)
g.sort()
print("kernels %d, span %.2f ms, busy %.2f ms (%.1f %%), idle %.2f ms in %d gaps (median %.2f us, > 5 us: %d totalling %.2f ms, > 20 us: %d)"
      % (len(step), span / 1e6, busy / 1e6, 100.0 * busy / span, sum(g) / 1e6, len(g), (g[len(g) // 2] if g else 0) / 1e3,
         sum(1 for x in g if x > 5000), sum(x for x in g if x > 5000) / 1e6, sum(1 for x in g if x > 20000)))
d = sorted(e - s for s, e, _ in step)
print("kernel durations: median %.1f us, p90 %.1f us, < 10 us: %d kernels" % (d[len(d) // 2] / 1e3, d[int(len(d) * 0.9)] / 1e3, sum(1 for x in d if x < 10000)))
