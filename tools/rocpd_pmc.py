#!/usr/bin/env python
"""Per-kernel averages of one PMC counter from a rocprofv3 rocpd database (--kernel-trace --pmc <COUNTER>).
    python tools/rocpd_pmc.py <db> [name-substring]   -> name, launches, avg counter value, avg duration"""
import sqlite3
import sys

db, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
c = sqlite3.connect(db)
agg = {}
for name, cn, val, dur in c.execute("select name, counter_name, counter_value, duration from pmc_events"):
    if sub and sub not in name:
        continue
    a = agg.setdefault((name, cn), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += val
    a[2] += dur
print("| kernel | counter | launches | avg value | avg duration us |")
print("|---|---|---|---|---|")
for (name, cn), a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print("| `%s` | %s | %d | %.1f | %.1f |" % (name[:90], cn, a[0], a[1] / a[0], a[2] / a[0] / 1e3))
