#!/usr/bin/env python
"""Per-kernel mean of PMC counters from a rocprofv3 --pmc rocpd database, grouped by (kernel, grid size) so
that the C3-sized dispatches are not averaged with the tiny start-up launches of runtime.warm_start()."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, grid_size, counter_name, value from counters_collection").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name, grid, counter, value in rows:
    if "gsr::" in name:
        agg[(name, int(grid))][counter].append(float(value))
for (name, grid), d in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    print(f"{name[:70]}  grid={grid}")
    for c, v in sorted(d.items()):
        print(f"    {c:<28} mean {sum(v) / len(v):>16.1f}  n={len(v)}")
