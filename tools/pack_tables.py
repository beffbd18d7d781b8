#!/usr/bin/env python3
"""Build the RRTMG data blobs shipped with climt_amd (run in the build container only).

Tables are DATA, not code.  This tool reads the numerical tables of the reference RRTMG
(k-distributions, cloud/aerosol optical-property tables, reference atmosphere, Planck
tables, g-point reduction bookkeeping) out of the *compiled reference library* built by
oracle/build_ref.sh -- i.e. the module variables after rrtmg_{sw,lw}_ini has run -- and
writes them in a neutral container (format below).  Array names/extents are discovered by
parsing the declaration lines of the reference's data modules (rrsw_kgNN.f90, rrsw_cld.f90,
rrsw_aer.f90, rrsw_ref.f90, rrsw_wvn.f90, and the rrlw_* equivalents) at pack time.

  SW blob  : RAW 16-g tables (kao, kbo, selfrefo, ... from rrtmg_sw_k_g.f90) + small tables.
             The 224->112 g-point reduction is done by the product at init (csrc/tables.cpp).
  LW blob  : RAW 16-g tables as the compiled reference library holds them after rrtmg_lw_ini, plus
             every in-tree small table (Planck totplnk, chi_mls, cloud tables, reduction
             bookkeeping).  Where the raw tables come from depends on how oracle/build_ref.sh linked
             the library (oracle/_ref/lw_kdata.txt):
               "file ..." the reference's data file rrtmg_lw_k_g.f90 was compiled in: its loaders
                          lw_kgb01..16 filled the module arrays -> "lw/meta/synthetic" = 0;
               "stub"     the file is a missing blob (this checkout): empty loaders, the module arrays
                          are filled with SYNTHETIC tables (tools/synth_lw_tables.py) before
                          rrtmg_lw_ini -> "lw/meta/synthetic" = 1.
             tools/ingest_lw_data.sh <rrtmg_lw_k_g.f90> runs the whole chain for a real file.
             --from-nc <rrtmg_lw.nc>: the data in AER's netCDF layout instead (what rrtmg_lw_read_nc.f90 reads;
             tools/lw_netcdf.py): the raw tables are read from the file and written into the module arrays of the
             stub-linked library before rrtmg_lw_ini, as the synthetic ones are -> "lw/meta/synthetic" = 0.

  python tools/pack_tables.py [sw|lw|all] [--out <blob>] [--fixture-out <npz>] [--from-nc <rrtmg_lw.nc>]   (defaults: the shipped paths)

Container: magic "RRTBL001", u32 count, then per entry
   u32 namelen, name, u32 dtype(0=f64,1=i32), u32 ndim, u32 dims[ndim] (Fortran order,
   first index fastest), u64 nbytes, payload padded to 8 bytes.

Also writes tests/golden/{sw,lw}_reduced_tables.npz: the reference's own post-init REDUCED
tables, used by tests to check the product's / oracle's g-point reduction.
"""
import os
import re
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("CLIMT_REFERENCE", "/root/reference")


def parse_params(path):
    par = {}
    for line in open(path):
        line = line.split("!")[0]
        m = re.search(r"parameter\s*::\s*(\w+)\s*=\s*([-+0-9.eE_a-z]+)", line, re.I)
        if m:
            v = m.group(2).lower().replace("_rb", "")
            try:
                par[m.group(1).lower()] = int(v)
            except ValueError:
                par[m.group(1).lower()] = float(v)
    return par


def _dim(expr, par):
    expr = expr.strip().lower()
    if ":" in expr:
        lo, hi = expr.split(":")
        return _dim(hi, par) - _dim(lo, par) + 1
    return int(eval(expr, {}, par))


def parse_module(path, par):
    """-> (module_name, [(name, dtype, dims)]) for non-parameter real/integer variables."""
    par = dict(par)
    par.update(parse_params(path))
    txt = []
    cur = ""
    for raw in open(path):
        line = raw.split("!")[0].rstrip()
        if not line.strip():
            continue
        if line.rstrip().endswith("&"):
            cur += line.rstrip()[:-1]
            continue
        txt.append(cur + line)
        cur = ""
    mod = None
    out = []
    for line in txt:
        m = re.match(r"\s*module\s+(\w+)", line, re.I)
        if m and mod is None:
            mod = m.group(1).lower()
        m = re.match(r"\s*(real|integer)\s*\(kind=\w+\)\s*(.*?)::\s*(.*)$", line, re.I)
        if not m or "parameter" in m.group(2).lower():
            continue
        dtype = np.float64 if m.group(1).lower() == "real" else np.int32
        attr = m.group(2)
        dm = re.search(r"dimension\s*\(([^)]*)\)", attr, re.I)
        common = [_dim(d, par) for d in dm.group(1).split(",")] if dm else None
        # split declarators at top-level commas
        decls, depth, tok = [], 0, ""
        for ch in m.group(3):
            if ch == "(":
                depth += 1
            if ch == ")":
                depth -= 1
            if ch == "," and depth == 0:
                decls.append(tok)
                tok = ""
            else:
                tok += ch
        decls.append(tok)
        for d in decls:
            d = d.strip()
            mm = re.match(r"(\w+)\s*(\((.*)\))?", d)
            name = mm.group(1).lower()
            dims = [_dim(x, par) for x in mm.group(3).split(",")] if mm.group(3) else (common or [])
            out.append((name, dtype, dims))
    return mod, out


class Blob:
    def __init__(self):
        self.entries = []

    def add(self, name, arr, dims=None):
        arr = np.asarray(arr)
        if arr.dtype.kind == "f":
            arr, code = arr.astype(np.float64), 0
        else:
            arr, code = arr.astype(np.int32), 1
        dims = list(arr.shape) if dims is None else list(dims)
        data = np.asfortranarray(arr).ravel(order="F")
        self.entries.append((name, code, dims, data))

    def write(self, path):
        with open(path, "wb") as f:
            f.write(b"RRTBL001")
            f.write(struct.pack("<I", len(self.entries)))
            for name, code, dims, data in self.entries:
                nb = name.encode()
                f.write(struct.pack("<I", len(nb)))
                f.write(nb)
                f.write(struct.pack("<II", code, len(dims)))
                for d in dims:
                    f.write(struct.pack("<I", d))
                payload = data.tobytes()
                f.write(struct.pack("<Q", len(payload)))
                f.write(payload)
                f.write(b"\0" * ((-len(payload)) % 8))
                pos = f.tell()
                f.write(b"\0" * ((-pos) % 8))


def read_blob(path):
    out = {}
    with open(path, "rb") as f:
        assert f.read(8) == b"RRTBL001"
        (n,) = struct.unpack("<I", f.read(4))
        for _ in range(n):
            (ln,) = struct.unpack("<I", f.read(4))
            name = f.read(ln).decode()
            code, nd = struct.unpack("<II", f.read(8))
            dims = struct.unpack("<%dI" % nd, f.read(4 * nd)) if nd else ()
            (nbytes,) = struct.unpack("<Q", f.read(8))
            dt = np.float64 if code == 0 else np.int32
            data = np.frombuffer(f.read(nbytes), dtype=dt)
            f.read((-nbytes) % 8)
            f.read((-f.tell()) % 8)
            out[name] = data.reshape(dims, order="F") if nd else data.reshape(())
    return out


RAW_SUFFIX = "o"


def dump_modules(ref, libdir, prefix, modfiles, par, skip_reduced=True):
    """-> (raw_and_small: dict name->array, reduced: dict)"""
    keep, reduced = {}, {}
    for mf in modfiles:
        mod, decls = parse_module(os.path.join(libdir, mf), par)
        names = {d[0] for d in decls}
        for name, dtype, dims in decls:
            try:
                a = ref.module_array(mod, name, tuple(dims) if dims else (1,), dtype)
            except ValueError:
                continue  # equivalenced alias (ka/kb...) or not exported
            a = np.array(a, copy=True)
            key = "%s/%s/%s" % (prefix, mod.replace("rrsw_", "").replace("rrlw_", ""), name)
            is_kg = "_kg" in mod
            if is_kg:
                # kg modules: NAMEo = raw 16-g table, NAME = reduced table built at init;
                # absa/absb (equivalenced with ka/kb) and ka_m*/kb_m* are reduced as well.
                is_red = (name + "o") in names or name in ("absa", "absb") or bool(re.match(r"k[ab]_m", name))
                (reduced if is_red else keep)[key] = (a, dims)
            else:
                keep[key] = (a, dims)
    return keep, reduced


def pack_sw():
    from oracle.ref_driver import RefSW
    libdir = os.path.join(REF, "climt/_lib/rrtmg_sw")
    par = parse_params(os.path.join(libdir, "parrrsw.f90"))
    ref = RefSW()
    ref.init()
    mods = ["rrsw_kg%d.f90" % b for b in range(16, 30)] + ["rrsw_cld.f90", "rrsw_aer.f90", "rrsw_ref.f90", "rrsw_wvn.f90"]
    keep, reduced = dump_modules(ref, libdir, "sw", mods, par)
    blob = Blob()
    for k, (a, dims) in sorted(keep.items()):
        if k.endswith("wvn/rwgt"):
            reduced[k] = (a, dims)      # derived at init by the product
            continue
        blob.add(k, a, dims)
    # NRLSSI2 mean-solar-cycle index tables (isolvar = 1): array constructors local to inatm_sw
    src = open(os.path.join(libdir, "rrtmg_sw_rad.nomcica.f90")).read()
    for name in ("mgavgcyc", "sbavgcyc"):
        m = re.search(name + r"\(:\)\s*=\s*\(/(.*?)/\)", src, re.S)
        vals = np.array([float(x) for x in re.findall(r"([0-9]+\.[0-9]+)_rb", m.group(1))])
        assert vals.size == 132, (name, vals.size)
        blob.add("sw/sol/" + name, vals, (132,))
    blob.add("sw/meta/synthetic", np.array([0], dtype=np.int32))
    out = os.path.join(ROOT, "climt_amd", "data", "rrtmg_sw_data.bin")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    blob.write(out)
    red = {k.replace("/", "__"): a for k, (a, d) in reduced.items()}
    red["sw__tbl__exp_tbl"] = np.array(ref.module_array("rrsw_tbl", "exp_tbl", (10001,)))
    red["sw__con__heatfac"] = np.array(ref.module_scalar("rrsw_con", "heatfac"))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sw_reduced_tables.npz"), **red)
    print("SW blob:", out, os.path.getsize(out), "bytes;", len(blob.entries), "entries; reduced fixture:", len(red))


def pack_lw(out=None, fixture_out=None, from_nc=None):
    from oracle import ref_driver
    from oracle.ref_driver import RefLW
    from tools.synth_lw_tables import fill_reference_modules
    libdir = os.path.join(REF, "climt/_lib/rrtmg_lw")
    par = parse_params(os.path.join(libdir, "parrrtm.f90"))
    ref = RefLW()
    kdata = ref_driver.lw_kdata()
    if from_nc:
        if kdata != "stub":
            raise SystemExit("pack_lw --from-nc: the reference library already has a data file compiled in (%s)" % kdata)
        from tools.lw_netcdf import read_lw_netcdf
        raw = read_lw_netcdf(from_nc)
        unset = sum(int(np.isnan(a).sum()) for a in raw.values())
        if unset:
            raise SystemExit("pack_lw --from-nc: %d table values were not found in %s" % (unset, from_nc))

        def fill(r):
            for (b, name), arr in raw.items():
                r.module_array("rrlw_kg%02d" % b, name, arr.shape)[...] = arr
        ref.init(fill_tables=fill)
        kdata = "netcdf " + from_nc
    elif kdata == "stub":
        ref.init(fill_tables=lambda r: fill_reference_modules(r, libdir, par))
    else:
        ref.init()      # the library's own loaders (the reference's data file) fill the raw tables
    mods = ["rrlw_kg%02d.f90" % b for b in range(1, 17)] + ["rrlw_cld.f90", "rrlw_ref.f90", "rrlw_wvn.f90"]
    keep, reduced = dump_modules(ref, libdir, "lw", mods, par)
    blob = Blob()
    for k, (a, dims) in sorted(keep.items()):
        if k.endswith("wvn/rwgt"):
            reduced[k] = (a, dims)
            continue
        blob.add(k, a, dims)
    if kdata != "stub":
        kao = keep["lw/kg01/kao"][0]
        if not np.any(kao != 0.0):
            raise SystemExit("pack_lw: the library says its k-data come from '%s' but lw/kg01/kao is all zero" % kdata)
    blob.add("lw/meta/synthetic", np.array([1 if kdata == "stub" else 0], dtype=np.int32))
    out = out or os.path.join(ROOT, "climt_amd", "data", "rrtmg_lw_data.bin")
    blob.write(out)
    red = {k.replace("/", "__"): a for k, (a, d) in reduced.items()}
    for t in ("exp_tbl", "tau_tbl", "tfn_tbl"):
        red["lw__tbl__" + t] = np.array(ref.module_array("rrlw_tbl", t, (10001,)))
    red["lw__con__heatfac"] = np.array(ref.module_scalar("rrlw_con", "heatfac"))
    fixture_out = fixture_out or os.path.join(ROOT, "tests", "golden", "lw_reduced_tables.npz")
    np.savez_compressed(fixture_out, **red)
    print("LW blob:", out, os.path.getsize(out), "bytes;", len(blob.entries), "entries; reduced fixture:", len(red),
          "; k-data:", kdata, "-> synthetic =", 1 if kdata == "stub" else 0)


if __name__ == "__main__":
    argv = sys.argv[1:]
    opts = {}
    for flag in ("--out", "--fixture-out", "--from-nc"):
        if flag in argv:
            i = argv.index(flag)
            opts[flag] = argv[i + 1]
            del argv[i:i + 2]
    what = argv[0] if argv else "all"
    if opts and what == "all":
        raise SystemExit("--out / --fixture-out need an explicit 'sw' or 'lw'")
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    if what in ("sw", "all"):
        if opts:
            raise SystemExit("--out / --fixture-out are implemented for 'lw' only")
        pack_sw()
    if what in ("lw", "all"):
        pack_lw(opts.get("--out"), opts.get("--fixture-out"), opts.get("--from-nc"))
