#!/usr/bin/env python3
"""Per-basic-block summary of a kernel's device assembly (hipcc -S --cuda-device-only): instruction count, MFMAs, scratch
accesses, lane spills (v_readlane / v_writelane), barriers, global loads, waits -- to see WHERE a kernel spills or waits.
    python tools/isa_blocks.py file.s kernel_name_substring"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^\S*%s\S*:" % re.escape(name), l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))      # (a kernel may hold several s_endpgm)
blk, order, stats = "entry", ["entry"], {"entry": dict(n=0, mfma=0, scratch=0, lane=0, bar=0, gload=0, vm0=0, line=start)}
for i in range(start + 1, end + 1):
    l = lines[i].strip()
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blk = m.group(1)
        order.append(blk)
        stats[blk] = dict(n=0, mfma=0, scratch=0, lane=0, bar=0, gload=0, vm0=0, line=i)
        continue
    if not l or l.startswith(";") or l.startswith("."):
        continue
    s = stats[blk]
    s["n"] += 1
    s["mfma"] += "v_mfma" in l
    s["scratch"] += "scratch_" in l
    s["lane"] += ("v_readlane" in l) or ("v_writelane" in l)
    s["bar"] += "s_barrier" in l
    s["gload"] += "global_load" in l
    s["vm0"] += bool(re.search(r"s_waitcnt.*vmcnt\(0\)", l))
for b in order:
    s = stats[b]
    if s["mfma"] or s["scratch"] or s["lane"] > 4 or s["n"] > 150:
        print(b, s)
