#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database: per-kernel count / total / average duration.
usage: tools/rocpd_stats.py results.db [> profiles/xxx_kernel_stats.txt]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    short = re.sub(r"\(.*$", "", name)
    short = re.sub(r"^void ", "", short)
    a = agg.setdefault(short, [0, 0.0, 1e30, 0.0])
    d = (e - s) / 1e3
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print("%-64s %6s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-64s %6d %12.1f %10.1f %10.1f %10.1f %6.2f" % (k[:64], a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
print("TOTAL kernel time us: %.1f" % tot)
