#!/usr/bin/env python3
"""Timeline excerpt of a rocprofv3 rocpd database (development tool): the last N kernel dispatches with start / end in
microseconds relative to the first of them -- to see which kernels overlap.   usage: tools/rocpd_timeline.py results.db [N=40]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = db.execute("select name, start, end from kernels order by start").fetchall()[-n:]
t0 = rows[0][1]
for name, s, e in rows:
    short = re.sub(r"^void ", "", re.sub(r"\(.*$", "", name))[:52]
    print("%-52s %10.1f %10.1f  (%8.1f us)" % (short, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
