#!/usr/bin/env python
"""Per-kernel summary (calls, total / mean / min / max duration) of a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on ROCm 7.2).
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# {'kernel':<72} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) <= 72 else name[:69] + "..."
        print(f"  {short:<72} {a[0]:>6} {a[1] / 1e3:>12.1f} {a[1] / a[0] / 1e3:>10.2f} {a[2] / 1e3:>10.2f} {a[3] / 1e3:>10.2f} {100 * a[1] / total:>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
