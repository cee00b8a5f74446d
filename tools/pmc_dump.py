#!/usr/bin/env python
"""Per-kernel mean of PMC counters from a rocprofv3 --pmc rocpd database, grouped by (kernel, grid size) so
that the C3-sized dispatches are not averaged with the tiny start-up launches of runtime.warm_start().
Dispatches whose value is below 2 % of the group's maximum are dropped (`skipped`): they are the speculative launches
that left at their first instruction because the binning capacity was too small (gsr_api.hip forward_impl)."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, grid_size, counter_name, value from counters_collection").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name, grid, counter, value in rows:
    if "gsr::" in name or "calib_" in name:
        agg[(name, int(grid))][counter].append(float(value))
for (name, grid), d in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    print(f"{name[:70]}  grid={grid}")
    for c, v in sorted(d.items()):
        top = max(v)
        keep = [x for x in v if x >= 0.02 * top] if top > 0 else v
        skipped = len(v) - len(keep)
        print(f"    {c:<28} mean {sum(keep) / len(keep):>16.1f}  n={len(keep)}" + (f"  skipped={skipped}" if skipped else ""))
