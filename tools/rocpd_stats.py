#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel stats table (the same columns as
rocprofv3's kernel_stats.csv): name, calls, total / average / min / max duration, share of GPU kernel time.
    python tools/rocpd_stats.py gpurun_out/prof/*/*_results.db [--skip-first-ms X] > profiles/rNN_kernel_stats.md"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cur = c.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        d = (e - s) / 1e3
        a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    span = (max(r[2] for r in rows) - min(r[1] for r in rows)) / 1e3
    print("| kernel | calls | total us | avg us | min us | max us | % of kernel time |")
    print("|---|---|---|---|---|---|---|")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = n if len(n) < 110 else n[:107] + "..."
        print("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.2f |" % (short, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
    print("\ntotal kernel time %.1f us over %d dispatches; first-to-last dispatch span %.1f us (GPU busy %.1f %%)" % (
        tot, len(rows), span, 100 * tot / span))


if __name__ == "__main__":
    main()
