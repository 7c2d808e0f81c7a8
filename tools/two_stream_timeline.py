#!/usr/bin/env python3
"""Two-stream timeline from a rocprofv3 --kernel-trace CSV of bench.py (fp32 mode: weight-gradient kernels on a second stream,
bmc_hip.ops.wgrad_side).  A backward pass = one burst of kernels on the weight-gradient stream (the stream with fewer kernels);
per burst: its span, how long both streams / only the main stream / only the weight-gradient stream / neither had a kernel
running, and how long the weight-gradient stream runs on after the main stream's last kernel of the pass (= what the join at
the end of backward waits for).       python tools/two_stream_timeline.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

bys = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    bys[r["Stream_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
order = sorted(bys, key=lambda q: -len(bys[q]))
for q in order:
    print("stream %s: %d kernels, busy %.1f ms" % (q, len(bys[q]), sum(e - s for s, e, _ in bys[q]) / 1e6))
if len(order) < 2:
    sys.exit("one stream only")
main, side = sorted(bys[order[0]]), sorted(bys[order[1]])
bursts, cur = [], [side[0]]
for k in side[1:]:
    if k[0] - max(e for _, e, _ in cur[-4:]) > 3_000_000:
        bursts.append(cur); cur = []
    cur.append(k)
bursts.append(cur)
for bi, b in enumerate(bursts):
    s0, s1 = b[0][0], max(e for _, e, _ in b)
    ev = []
    for s, e, _ in main:
        if e > s0 and s < s1:
            ev.append((max(s, s0), 1, 0)); ev.append((min(e, s1), -1, 0))
    for s, e, _ in b:
        ev.append((s, 0, 1)); ev.append((e, 0, -1))
    ev.sort()
    m = sd = 0
    acc = [0, 0, 0, 0]
    last = s0
    for t, dm, ds in ev:
        acc[(1 if m else 0) + (2 if sd else 0)] += t - last
        m += dm; sd += ds; last = t
    after = [k for k in main if k[0] >= s0 and k[0] < s1]
    mlast = max(e for _, e, _ in after) if after else s0
    tail = [k for k in b if k[1] > mlast]
    print("backward pass %d: %d weight-gradient kernels over %.1f ms: both %.1f, main only %.1f, weight-gradient stream only %.1f, neither %.1f ms; "
          "weight-gradient stream runs %.2f ms past the main stream's last kernel of the pass (%d kernels)"
          % (bi, len(b), (s1 - s0) / 1e6, acc[3] / 1e6, acc[1] / 1e6, acc[2] / 1e6, acc[0] / 1e6, max(0, s1 - mlast) / 1e6, len(tail)))
