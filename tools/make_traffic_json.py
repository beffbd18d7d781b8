#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the PMC passes of tools/collect_profiles.sh (profiles/<round>_pmc_{clear,cloudy}_{FETCH,WRITE}_SIZE.txt):
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE doubled per the gfx950 correction of
MI355X_MICROARCH.md (HBM section).  Key: "<kernel>|<columns>|<levels>|<clear|cloudy>" (what bench.py looks up).
usage: tools/make_traffic_json.py [round=r01] [columns=8192] [levels=60]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
nlev = int(sys.argv[3]) if len(sys.argv) > 3 else 60


def get(mode, counter):
    out = {}
    for line in open(os.path.join(ROOT, "profiles", "%s_pmc_%s_%s.txt" % (rnd, mode, counter))):
        m = re.match(r"(.*?)\s+%s\s+dispatches\s+\d+\s+avg\s+(\S+)" % counter, line.strip())
        if m and "_solve_" in m.group(1):
            out[m.group(1).strip()] = float(m.group(2))
    return out


traffic = {"_doc": __doc__.split("\n")[0] + " See tools/make_traffic_json.py."}
for mode in ("clear", "cloudy"):
    fetch, write = get(mode, "FETCH_SIZE"), get(mode, "WRITE_SIZE")
    for k in fetch:
        b = (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0
        if b > 1.0e6:
            traffic["%s|%d|%d|%s" % (k, ncol, nlev, mode)] = b
json.dump(traffic, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
