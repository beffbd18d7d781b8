#!/usr/bin/env python3
"""Write a `rrtmg_lw_k_g.f90` -- the longwave k-distribution data file of RRTMG_LW -- from raw 16-g tables.

TEST INFRASTRUCTURE for the data-ingestion path (tools/ingest_lw_data.sh, tests/test_lw_ingest.py).  The reference
checkout lacks this file (/root/reference/.MISSING_LARGE_BLOBS:3); the day a user has it, the ingestion path compiles
it into oracle/_ref, packs what the reference loaded and switches the longwave parity tests on.  To prove that path
WITHOUT the file, this tool writes a stand-in in the same syntax the reference's data files use (the shortwave one,
climt/_lib/rrtmg_sw/rrtmg_sw_k_g.f90, is present): one `subroutine lw_kgbNN` per band (the loaders rrtmg_lw_ini calls,
rrtmg_lw_init.f90:80-95), `use rrlw_kgNN, only : ...` of the raw arrays the module declares (rrlw_kgNN.f90), and one
array-constructor assignment per first-index slice, `kao(:, jt, jp, ig) = (/ ... /)`.  Values are printed with 17
significant digits so that a correctly rounding compiler reads back the same doubles.

  python tools/write_lw_k_g.py <out.f90>              # from the shipped blob's raw tables (synthetic today)
  python tools/write_lw_k_g.py <out.f90> <blob.bin>   # from another packed blob
"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def raw_tables_of_blob(blob):
    """{(band, name): array} of the raw 16-g arrays in a packed LW blob (entries lw/kgNN/<name>)."""
    out = {}
    for key, arr in blob.items():
        parts = key.split("/")
        if len(parts) == 3 and parts[0] == "lw" and parts[1].startswith("kg") and arr.ndim >= 1:
            out[(int(parts[1][2:]), parts[2])] = np.asarray(arr, dtype=np.float64)
    return out


def lower_bounds(libdir):
    """{(band, name): lower bound of every dimension} from the declarations of rrlw_kgNN.f90 (the upper-atmosphere
    tables are declared on reference-pressure levels 13:59, e.g. `kbo(5,13:59,no1)`, rrlw_kg01.f90:33)."""
    import re
    out = {}
    for b in range(1, 17):
        for line in open(os.path.join(libdir, "rrlw_kg%02d.f90" % b)):
            line = line.split("!")[0]
            m = re.match(r"\s*real\s*\(kind=\w+\)\s*::\s*(.*)$", line, re.I)
            if not m:
                continue
            for name, spec in re.findall(r"(\w+)\s*\(([^)]*)\)", m.group(1)):
                out[(b, name.lower())] = [int(d.split(":")[0]) if ":" in d else 1 for d in spec.split(",")]
    return out


def _constructor(values, indent="        "):
    lines = []
    vals = ["%.17e_rb" % v for v in values]
    for i in range(0, len(vals), 4):
        lines.append(indent + "& " + ",".join(vals[i:i + 4]) + ("," if i + 4 < len(vals) else "") + " &")
    lines[-1] = lines[-1][:-2] + " /)"
    return lines


def write_k_g(path, tables, lbounds=None):
    lbounds = lbounds or {}
    bands = sorted({b for b, _ in tables})
    with open(path, "w") as f:
        f.write("! rrtmg_lw_k_g.f90 stand-in written by tools/write_lw_k_g.py: %d raw tables, %d values.\n"
                % (len(tables), sum(a.size for a in tables.values())))
        f.write("! NOT AER data unless the tables it was written from were.\n")
        for b in bands:
            names = sorted(n for bb, n in tables if bb == b)
            f.write("      subroutine lw_kgb%02d\n\n" % b)
            f.write("      use parkind, only : im => kind_im, rb => kind_rb\n")
            f.write("      use rrlw_kg%02d, only : %s\n\n" % (b, ", &\n                            ".join(
                ", ".join(names[i:i + 6]) for i in range(0, len(names), 6))))
            f.write("      implicit none\n      save\n\n")
            for n in names:
                a = tables[(b, n)]
                lb = lbounds.get((b, n), [1] * a.ndim)
                if a.ndim == 1:
                    f.write("      %s(:) = (/ &\n" % n)
                    f.write("\n".join(_constructor(a)) + "\n")
                    continue
                for idx in itertools.product(*[range(d) for d in reversed(a.shape[1:])]):
                    idx = tuple(reversed(idx))          # Fortran order: the second index varies fastest
                    f.write("      %s(:,%s) = (/ &\n" % (n, ",".join("%d" % (i + l) for i, l in zip(idx, lb[1:]))))
                    f.write("\n".join(_constructor(a[(slice(None),) + idx])) + "\n")
            f.write("\n      end subroutine lw_kgb%02d\n\n" % b)


if __name__ == "__main__":
    from tools.pack_tables import read_blob
    out = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "climt_amd", "data", "rrtmg_lw_data.bin")
    tables = raw_tables_of_blob(read_blob(src))
    tables.pop((2, "refparam"), None)   # declared by rrlw_kg02 but set and used nowhere
    ref = os.environ.get("CLIMT_REFERENCE", "/root/reference")
    write_k_g(out, tables, lower_bounds(os.path.join(ref, "climt/_lib/rrtmg_lw")))
    print("wrote %s: %d tables, %.1f MB" % (out, len(tables), os.path.getsize(out) / 1e6))
