#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc run (rocpd sqlite .db) per kernel: mean duration and mean counter values per dispatch.
python tools/pmc_agg.py <results.db> [kernel-name substring ...]"""
import collections
import json
import sqlite3
import sys


def aggregate(db, filters=()):
    c = sqlite3.connect(db)
    sfx = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    q = f"""select d.id, k.kernel_name, d.grid_size_x, d.start, d.end, p.name, sum(e.value)
            from rocpd_kernel_dispatch{sfx} d join rocpd_info_kernel_symbol{sfx} k on d.kernel_id = k.id
            join rocpd_pmc_event{sfx} e on e.event_id = d.event_id join rocpd_info_pmc{sfx} p on p.id = e.pmc_id
            group by d.id, p.name order by d.id"""
    agg = collections.defaultdict(lambda: {"n": set(), "dur_us": {}, "ctr": collections.defaultdict(list)})
    for did, name, grid, st, en, pn, val in c.execute(q):
        if filters and not any(f in name for f in filters):
            continue
        a = agg[(name.split("(")[0][:70], grid)]
        a["n"].add(did)
        a["dur_us"][did] = (en - st) / 1e3
        a["ctr"][pn].append(val)
    out = []
    for (name, grid), a in agg.items():
        row = {"kernel": name, "grid": grid, "dispatches": len(a["n"]), "avg_us": sum(a["dur_us"].values()) / len(a["dur_us"])}
        for pn, v in a["ctr"].items():
            row[pn] = sum(v) / len(v)
        out.append(row)
    return out


if __name__ == "__main__":
    for r in aggregate(sys.argv[1], sys.argv[2:]):
        print(json.dumps(r))
