#!/usr/bin/env python
"""Per-kernel mean of PMC counters from a rocprofv3 --pmc rocpd database."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
rows = db.execute("select * from counters_collection").fetchall()
ci = {c: i for i, c in enumerate(cols)}
name_c = "kernel_name" if "kernel_name" in ci else [c for c in cols if "kernel" in c and "name" in c][0]
cn = "counter_name" if "counter_name" in ci else "name"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r[ci[name_c]]][r[ci[cn]]].append(float(r[ci["value"]]))
for k, d in agg.items():
    if "gsr::" not in k: continue
    print(k[:60])
    for c, v in sorted(d.items()):
        print(f"    {c:<28} mean {sum(v)/len(v):>16.1f}  n={len(v)}")
