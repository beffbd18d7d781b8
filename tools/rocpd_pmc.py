#!/usr/bin/env python3
"""Per-kernel PMC counter averages from a rocprofv3 rocpd SQLite database.
usage: tools/rocpd_pmc.py results.db [kernel-substring]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
namec = "kernel_name" if "kernel_name" in ix else "name"
agg = {}
for r in rows:
    kn = re.sub(r"\(.*$", "", str(r[ix[namec]])).replace("void ", "")
    if flt not in kn:
        continue
    cn = r[ix["counter_name"]]
    v = float(r[ix["value"]])
    did = r[ix["dispatch_id"]] if "dispatch_id" in ix else None
    a = agg.setdefault((kn, cn), {})
    a[did] = a.get(did, 0.0) + v          # sum over dimension instances of one dispatch
for (kn, cn), per in sorted(agg.items()):
    vals = list(per.values())
    print("%-40s %-24s dispatches %3d  avg %.6g" % (kn[:40], cn, len(vals), sum(vals) / len(vals)))
