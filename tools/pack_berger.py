#!/usr/bin/env python3
"""Pack the Berger (1978) orbital-series coefficient tables of climt's BergerSolarInsolation into
climt_amd/data/berger_tables.npz (run in the build container only).  The nine arrays (obliquity cosine series A, f,
delta; eccentricity series P, alpha, zeta; general-precession series F, f_prime, delta_prime) are module-level
`np.array([...])` literals in climt/_components/berger_solar_insolation.py:7-490; they are read with `ast` (data, no code)."""
import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CLIMT_REFERENCE", "/root/reference")
NAMES = ("A", "f", "delta", "P", "alpha", "zeta", "F", "f_prime", "delta_prime")


def main():
    tree = ast.parse(open(os.path.join(REF, "climt/_components/berger_solar_insolation.py")).read())
    out = {}
    for st in tree.body:
        if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name) and st.targets[0].id in NAMES:
            call = st.value                      # np.array([...])
            out[st.targets[0].id] = np.array(ast.literal_eval(call.args[0]), dtype=np.float64)
    missing = [n for n in NAMES if n not in out]
    if missing:
        sys.exit("tables not found: %s" % missing)
    assert (len(out["A"]), len(out["P"]), len(out["F"])) == (47, 19, 78)
    dst = os.path.join(ROOT, "climt_amd", "data", "berger_tables.npz")
    np.savez(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
