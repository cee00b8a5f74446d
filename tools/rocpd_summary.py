#!/usr/bin/env python
"""Per-kernel summary (calls, total / mean / min / max duration) of a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on ROCm 7.2).
Dispatches are grouped by (kernel, grid size) so that the benchmark-sized launches are not averaged with the
tiny start-up launches of runtime.warm_start().
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.txt"""
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


if __name__ == "__main__":
    main(sys.argv[1])
