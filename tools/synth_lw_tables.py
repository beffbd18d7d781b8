#!/usr/bin/env python3
"""SYNTHETIC longwave k-distribution tables (the reference's rrtmg_lw_k_g.f90 is a missing blob).

Every raw 16-g array of the reference modules rrlw_kg01..16 (shapes taken from the module
declarations) is filled with smooth, positive, deterministic numbers of physically plausible
magnitude so that all branches of taumol / rtrn (optically thin series, table look-ups, minor-gas
adjustments, Planck-fraction interpolation) are exercised.  They are NOT physical: longwave fluxes
computed from them only pin the ALGORITHM against the reference Fortran run on the same tables.
"""
import os
import zlib

import numpy as np

# rough per-band strength (log10 of the mid-g absorption coefficient of the key species)
_BAND_LOGK = {1: -1.5, 2: -2.0, 3: -2.2, 4: -1.0, 5: -1.8, 6: -3.0, 7: -2.6, 8: -3.0, 9: -2.4, 10: -1.8,
              11: -1.2, 12: -2.6, 13: -2.8, 14: 0.2, 15: -0.6, 16: -2.2}


def _rng(name):
    return np.random.default_rng(zlib.crc32(name.encode()))


def _gprofile(n=16):
    """k rises steeply across the g-points (sorted k-distribution)."""
    g = (np.arange(n) + 0.5) / n
    return 3.2 * (g - 0.5) + 1.8 * g ** 6


def synth_array(band, name, dims):
    rng = _rng("lw%02d/%s" % (band, name))
    dims = tuple(dims)
    gp = _gprofile()
    if name.startswith("fracref"):
        # (16,) or (16, n): positive, normalised over g for every mixture
        shape = dims if len(dims) > 1 else dims + (1,)
        w = np.exp(-0.5 * ((np.arange(16)[:, None] - 6.0 - 0.4 * np.arange(shape[1])[None, :]) / 5.0) ** 2)
        w = w * (1.0 + 0.2 * rng.uniform(-1, 1, shape))
        w = w / w.sum(axis=0, keepdims=True)
        return w.reshape(dims)
    if name in ("kao", "kbo"):
        # (..., 16) with leading (nsp?, 5 temperatures, pressures)
        lead = dims[:-1]
        npres = lead[-1]
        ntemp = lead[-2]
        nsp = lead[0] if len(lead) == 3 else 1
        base = _BAND_LOGK[band] + (0.4 if name == "kbo" else 0.0)
        p = np.arange(npres)[None, None, :, None]
        t = np.arange(ntemp)[None, :, None, None]
        s = np.arange(nsp)[:, None, None, None] / max(nsp - 1, 1)
        logk = base + gp[None, None, None, :] - 0.035 * p + 0.06 * (t - 2) + 0.5 * (s - 0.5) \
            + 0.05 * np.sin(0.7 * p + 1.3 * t + 2.1 * s)
        logk = logk + 0.02 * rng.uniform(-1, 1, logk.shape)
        return (10.0 ** logk).reshape(dims)
    if name == "selfrefo":      # (10, 16)
        t = np.arange(dims[0])[:, None]
        return 10.0 ** (_BAND_LOGK[band] - 0.2 + 0.6 * gp[None, :] - 0.03 * t) * (1 + 0.05 * rng.uniform(-1, 1, dims))
    if name == "forrefo":       # (4, 16)
        t = np.arange(dims[0])[:, None]
        return 10.0 ** (_BAND_LOGK[band] - 2.5 + 0.5 * gp[None, :] + 0.05 * t) * (1 + 0.05 * rng.uniform(-1, 1, dims))
    if name.startswith("kao_m") or name.startswith("kbo_m"):
        # (19,16) or (9|5,19,16): minor-gas coefficients
        lead = dims[:-1]
        t = np.arange(lead[-1])[:, None]
        v = 10.0 ** (-1.0 + 0.8 * gp[None, :] + 0.02 * t)
        if len(lead) == 2:
            s = np.arange(lead[0])[:, None, None] / max(lead[0] - 1, 1)
            v = v[None, :, :] * (1.0 + 0.6 * s)
        v = v * (1 + 0.05 * rng.uniform(-1, 1, v.shape))
        if "mn2" in name and "mn2o" not in name:
            v = v * 1.0e-5       # scaled by colbrd*scaleminor(n2) ~ 1e4..1e5
        if "mo2" in name:
            v = v * 1.0e-5
        return v.reshape(dims)
    if name in ("ccl4o", "cfc11adjo", "cfc12o", "cfc22adjo"):
        return 10.0 ** (3.0 + 0.5 * gp) * (1 + 0.05 * rng.uniform(-1, 1, dims))
    raise KeyError("no synthetic rule for %s (band %d, dims %s)" % (name, band, dims))


def raw_arrays(libdir, par):
    """-> {(band, name): array} for every raw table of rrlw_kg01..16."""
    from tools.pack_tables import parse_module
    out = {}
    for b in range(1, 17):
        mod, decls = parse_module(os.path.join(libdir, "rrlw_kg%02d.f90" % b), par)
        names = {d[0] for d in decls}
        for name, dtype, dims in decls:
            if not dims or dtype != np.float64:
                continue
            is_red = (name + "o") in names or name in ("absa", "absb", "ka", "kb") or name.startswith("ka_m") or name.startswith("kb_m")
            if is_red or name == "refparam":   # refparam (kg02) is declared but never used
                continue
            out[(b, name)] = synth_array(b, name, dims)
    return out


def fill_reference_modules(ref, libdir, par):
    """Write the synthetic raw tables into the reference library's rrlw_kgNN module arrays."""
    for (b, name), arr in raw_arrays(libdir, par).items():
        view = ref.module_array("rrlw_kg%02d" % b, name, arr.shape)
        view[...] = arr


def fill_reference_from_blob(ref, blob):
    """Same, from a packed LW blob (dict name -> array as returned by tools.pack_tables.read_blob)."""
    for key, arr in blob.items():
        parts = key.split("/")
        if len(parts) != 3 or not parts[1].startswith("kg") or arr.ndim == 0:
            continue
        try:
            view = ref.module_array("rrlw_" + parts[1], parts[2], arr.shape)
        except ValueError:
            continue
        view[...] = arr
