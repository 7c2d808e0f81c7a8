#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid size): count, mean / min / max duration (us)."""
import csv, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    rows[(name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), d in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print("%-48s grid %6d  n %5d  mean %8.1f  min %8.1f  max %8.1f  total %9.0f us" % (name, grid, len(d), sum(d) / len(d), min(d), max(d), sum(d)))
