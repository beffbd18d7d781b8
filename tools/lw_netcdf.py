#!/usr/bin/env python3
"""The longwave k-distribution data in AER's netCDF layout (`rrtmg_lw.nc`), the alternative to rrtmg_lw_k_g.f90.

RRTMG_LW ships its absorption coefficients either as the Fortran data file rrtmg_lw_k_g.f90 or as rrtmg_lw.nc, read by
climt/_lib/rrtmg_lw/rrtmg_lw_read_nc.f90 (the same sixteen loaders lw_kgb01..16, implemented with nf90_get_var).  Neither
file is in the reference checkout.  This module reads the netCDF file WITHOUT the netCDF Fortran library: the hyperslab
every loader reads into every raw table -- variable name, target array (or array section), start and count vectors, the
absorber index looked up by name -- is taken from the text of rrtmg_lw_read_nc.f90 at run time (build container only; the
parameters come from rrlw_ncpar.f90), and applied with scipy.io.netcdf_file (netCDF classic / 64-bit offset; a netCDF-4
file is converted with `nccopy -k classic`).

    raw = read_lw_netcdf("rrtmg_lw.nc")            # {(band, name): array in Fortran shape}, every raw table of rrlw_kg01..16
    python tools/pack_tables.py lw --from-nc rrtmg_lw.nc   # packs climt_amd/data/rrtmg_lw_data.bin from it (synthetic = 0)

write_lw_netcdf() is the inverse (a file in that layout from raw tables): test infrastructure for the reader
(tests/test_lw_ingest.py), since no real file is at hand.
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("CLIMT_REFERENCE", "/root/reference")
LIBDIR = os.path.join(REF, "climt/_lib/rrtmg_lw")


def _ncpar():
    """integer parameters and the absorber name list of rrlw_ncpar.f90"""
    txt = open(os.path.join(LIBDIR, "rrlw_ncpar.f90")).read()
    code = "\n".join(l.split("!")[0] for l in txt.split("\n"))
    par = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"(\w+)\s*=\s*(\d+)\s*[,&\n]", code)}
    names = [n.strip() for n in re.findall(r"'([A-Za-z0-9 ]+)'", code[code.index("AbsorberNames"):code.index("status")])]
    return par, names


def parse_read_nc():
    """-> {band: [(ncvar, target, section or None, start[], count[])]} with every symbol resolved to integers (1-based starts)."""
    par, absorbers = _ncpar()
    txt = open(os.path.join(LIBDIR, "rrtmg_lw_read_nc.f90")).read()
    # join continuation lines, drop comments
    lines, cur = [], ""
    for raw in txt.split("\n"):
        line = raw.split("!")[0].rstrip() if not raw.lstrip().startswith("!") else ""
        if line.rstrip().endswith("&"):
            cur += line.rstrip()[:-1] + " "
            continue
        lines.append(cur + line)
        cur = ""
    out, band, sym, ncvar = {}, None, {}, None
    for line in lines:
        m = re.match(r"\s*subroutine\s+lw_kgb(\d+)", line, re.I)
        if m:
            band = int(m.group(1))
            out[band] = []
            sym = dict(par)
            sym.update(bandnumber=band, gpointsetnumber=1, numgpoints=16)
            continue
        if band is None:
            continue
        m = re.search(r"parameter\s*::\s*(.*)$", line, re.I)
        if m:
            for n, v in re.findall(r"(\w+)\s*=\s*(\w+)", m.group(1)):
                v = v.lower()
                sym[n.lower()] = int(v) if v.isdigit() else (16 if re.fullmatch(r"no\d+", v) else sym.get(v, sym.get(n.lower())))
        m = re.search(r"getAbsorberIndex\(\s*'(\w+)'\s*,\s*(\w+)\s*\)", line)
        if m:
            sym[m.group(2).lower()] = absorbers.index(m.group(1).strip()) + 1
        m = re.search(r'nf90_inq_varid\(\s*ncid\s*,\s*"(\w+)"', line)
        if m:
            ncvar = m.group(1)
        m = re.search(r"nf90_get_var\(\s*ncid\s*,\s*varID\s*,\s*(\w+)\s*(\([^)]*\))?\s*,\s*start\s*=\s*\(/(.*?)/\)\s*,\s*count\s*=\s*\(/(.*?)/\)", line)
        if m:
            def vec(s):
                return [int(t) if t.strip().isdigit() else sym[t.strip().lower()] for t in s.split(",")]
            out[band].append((ncvar, m.group(1).lower(), m.group(2), vec(m.group(3)), vec(m.group(4))))
    return out


def _target_view(arr, section):
    """The Fortran array section `section` (e.g. "(:,1:5)") of arr (Fortran shape) as a writable view."""
    if not section:
        return arr
    idx = []
    for t in section.strip("()").split(","):
        t = t.strip()
        if t == ":":
            idx.append(slice(None))
        elif ":" in t:
            a, b = t.split(":")
            idx.append(slice(int(a) - 1, int(b)))
        else:
            idx.append(int(t) - 1)
    return arr[tuple(idx)]


def _raw_shapes():
    from tools.pack_tables import parse_module, parse_params
    par = parse_params(os.path.join(LIBDIR, "parrrtm.f90"))
    shapes = {}
    for b in range(1, 17):
        _, decls = parse_module(os.path.join(LIBDIR, "rrlw_kg%02d.f90" % b), par)
        for name, dtype, dims in decls:
            if dims and dtype == np.float64:
                shapes[(b, name)] = tuple(dims)
    return shapes


def read_lw_netcdf(path):
    """Every raw 16-g table the loaders of rrtmg_lw_read_nc.f90 fill: {(band, name): array (Fortran shape)}."""
    from scipy.io import netcdf_file
    plan, shapes = parse_read_nc(), _raw_shapes()
    nc = netcdf_file(path, "r", mmap=False)
    out = {}
    for band, recs in plan.items():
        for ncvar, target, section, start, count in recs:
            var = nc.variables[ncvar]
            sl = tuple(slice(s - 1, s - 1 + c) for s, c in zip(reversed(start), reversed(count)))      # C order = reversed Fortran order
            flat = np.asarray(var[sl], dtype=np.float64).ravel()                                      # the hyperslab in Fortran memory order
            arr = out.setdefault((band, target), np.full(shapes[(band, target)], np.nan))
            view = _target_view(arr, section)
            if view.size != flat.size:
                raise ValueError("band %d %s%s: the hyperslab has %d values, the target %d" % (band, target, section or "", flat.size, view.size))
            view[...] = flat.reshape(view.shape, order="F")
    nc.close()
    return out


# dimensions of the file's variables in Fortran order (rrlw_ncpar.f90 names), as the loaders' start / count vectors imply
_VAR_DIMS = {
    "PlanckFractionLowerAtmos": ("gpoint", "keylower", "band", "gpointset"),
    "PlanckFractionUpperAtmos": ("gpoint", "keyupper", "band", "gpointset"),
    "KeySpeciesAbsorptionCoefficientsLowerAtmos": ("keylower", "tdiff", "plower", "gpoint", "band", "gpointset"),
    "KeySpeciesAbsorptionCoefficientsUpperAtmos": ("keyupper", "tdiff", "pupper", "gpoint", "band", "gpointset"),
    "H20SelfAbsorptionCoefficients": ("tself", "gpoint", "band", "gpointset"),
    "H20ForeignAbsorptionCoefficients": ("tforeign", "gpoint", "band", "gpointset"),
    "AbsorptionCoefficientsLowerAtmos": ("keylower", "t", "gpoint", "absorber", "band", "gpointset"),
    "AbsorptionCoefficientsUpperAtmos": ("keyupper", "t", "gpoint", "absorber", "band", "gpointset"),
}


def write_lw_netcdf(path, tables):
    """A file in the layout read above, holding `tables` ({(band, name): array}) at the places the loaders read them from."""
    from scipy.io import netcdf_file
    par, _ = _ncpar()
    plan = parse_read_nc()
    nc = netcdf_file(path, "w", version=2)
    data = {}
    for v, dims in _VAR_DIMS.items():
        for dname in dims:
            if dname not in nc.dimensions:
                nc.createDimension(dname, par[dname])
        data[v] = np.zeros(tuple(par[dn] for dn in reversed(dims)))
    for band, recs in plan.items():
        for ncvar, target, section, start, count in recs:
            if (band, target) not in tables:
                continue
            view = _target_view(np.asarray(tables[(band, target)], dtype=np.float64), section)
            sl = tuple(slice(s - 1, s - 1 + c) for s, c in zip(reversed(start), reversed(count)))
            data[ncvar][sl] = view.ravel(order="F").reshape(data[ncvar][sl].shape)
    for v, dims in _VAR_DIMS.items():
        var = nc.createVariable(v, "d", tuple(reversed(dims)))
        var[:] = data[v]
    nc.close()


if __name__ == "__main__":
    raw = read_lw_netcdf(sys.argv[1])
    print("%d raw tables, %d values, unset: %d" % (len(raw), sum(a.size for a in raw.values()), sum(int(np.isnan(a).sum()) for a in raw.values())))
