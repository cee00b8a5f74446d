#!/usr/bin/env python
"""Per-kernel summary (calls, total / mean / min / max duration) of a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on ROCm 7.2).
Dispatches are grouped by (kernel, grid size) so that the benchmark-sized launches are not averaged with the
tiny start-up launches of runtime.warm_start().
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.txt
       python tools/rocpd_summary.py x_results.db --sequence 60   # the last 60 dispatches in launch order, with the idle gap before each"""
import os
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, lds_size from kernels").fetchall()
    groups = {}
    for name, s, e, gx, wx, vg, lds in rows:
        groups.setdefault((name, gx, wx, vg, lds), []).append(e - s)
    # a dispatch shorter than 2 % of its group's longest is a speculative launch that left at its first instruction
    # (binning capacity too small, gsr_api.hip forward_impl): counted separately, not averaged in
    agg = {}
    for k, ds in groups.items():
        top = max(ds)
        keep = [d for d in ds if d >= 0.02 * top]
        agg[k] = [len(keep), sum(keep), min(keep), max(keep), len(ds) - len(keep)]
    total = sum(a[1] for a in agg.values()) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {os.path.basename(os.path.dirname(path))}/{os.path.basename(path)}  (grouped by kernel and grid size; grid in work-items;")
    print("#  `skip` = speculative launches that exited at once, not averaged)")
    print(f"# {'kernel':<58} {'grid':>9} {'wg':>4} {'vgpr':>4} {'lds':>6} {'calls':>5} {'skip':>4} {'total_us':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}")
    for (name, gx, wx, vg, lds), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if a[1] / total < 0.0005:
            continue
        short = name if len(name) <= 58 else name[:55] + "..."
        print(f"  {short:<58} {gx:>9} {wx:>4} {vg:>4} {lds:>6} {a[0]:>5} {a[4]:>4} {a[1] / 1e3:>10.1f} {a[1] / a[0] / 1e3:>9.2f} {a[2] / 1e3:>9.2f} {a[3] / 1e3:>9.2f} {100 * a[1] / total:>6.2f}")


def sequence(path, n):
    """the last n dispatches (kernels and, when the database has them, memory copies) in start order"""
    db = sqlite3.connect(path)
    rows = [(s, e, name, gx) for name, s, e, gx in db.execute("select name, start, end, grid_x from kernels")]
    try:
        rows += [(s, e, "<memcpy " + str(nm) + ">", sz) for nm, s, e, sz in db.execute("select name, start, end, size from memory_copies")]
    except sqlite3.Error:
        pass
    rows.sort()
    rows = rows[-n:]
    prev = None
    print(f"# {'start_us':>10} {'gap_us':>7} {'dur_us':>8}  {'grid':>9}  kernel")
    t0 = rows[0][0]
    for s, e, name, gx in rows:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        print(f"  {(s - t0) / 1e3:>10.1f} {gap:>7.1f} {(e - s) / 1e3:>8.1f}  {gx:>9}  {name[:90]}")
        prev = max(prev, e) if prev is not None else e


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--sequence":
        sequence(sys.argv[1], int(sys.argv[3]))
    else:
        main(sys.argv[1])
