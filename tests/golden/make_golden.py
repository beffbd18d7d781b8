#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference checkout (run in the build container only).

1. climt_cache_<Class>-<descriptor>.npz -- (input state, expected output) pairs recovered from the
   reference's own golden caches tests/cached_component_output/TestRRTMG*-{column,3d}[_stepping]-{0,1}.cache
   (netCDF3).  `*_stepping-1.cache` is the full stepped state: every input array,
   with air_temperature advanced by 10 s x tendency (Euler first step of AdamsBashforth), so the
   un-stepped temperature is recovered as T1 - 10 s * tendency / 86400.
2. ref_sw_*.npz / ref_lw_*.npz -- outputs of the reference Fortran (oracle/_ref) on seeded synthetic
   columns (climt_amd.synthetic) for the boundary-level parity tests that must run where the
   reference library is not available.  LW fixtures are on the SYNTHETIC k-tables of the LW blob.
"""
import os
import sys

import numpy as np
from scipy.io import netcdf_file

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.environ.get("CLIMT_REFERENCE", "/root/reference")
CACHE = os.path.join(REF, "tests", "cached_component_output")
OUT = os.path.dirname(os.path.abspath(__file__))


def read_cache(fn):
    f = netcdf_file(os.path.join(CACHE, fn), mmap=False)
    out = {}
    for k, v in f.variables.items():
        units = getattr(v, "units", b"")
        units = units.decode("utf-8") if isinstance(units, bytes) else units
        out[k] = (np.array(v[...]), tuple(v.dimensions), units)
    return out


def cache_case(cls, desc):
    tend = read_cache("%s-%s-0.cache" % (cls, desc))
    diag = read_cache("%s-%s-1.cache" % (cls, desc))
    step = "%s-%s_stepping-1.cache" % (cls, desc)
    if not os.path.exists(os.path.join(CACHE, step)):
        return None
    state = read_cache(step)
    t1, dims, units = state["air_temperature"]
    dt = 10.0
    tt = tend["air_temperature"][0]
    # bring the tendency to the state's dim order
    tdims = tend["air_temperature"][1]
    tt = np.transpose(tt, [tdims.index(d) for d in dims])
    state["air_temperature"] = (t1 - dt * tt / 86400.0, dims, units)
    save = {}
    for k, (v, d, u) in state.items():
        if k == "time":
            continue
        save["state/%s/values" % k] = v
        save["state/%s/dims" % k] = np.array(",".join(d))
        save["state/%s/units" % k] = np.array(u)
    for grp, dd in (("tend", tend), ("diag", diag)):
        for k, (v, d, u) in dd.items():
            save["%s/%s/values" % (grp, k)] = v
            save["%s/%s/dims" % (grp, k)] = np.array(",".join(d))
            save["%s/%s/units" % (grp, k)] = np.array(u)
    np.savez_compressed(os.path.join(OUT, "climt_cache_%s-%s.npz" % (cls, desc)), **save)
    return len(save)


def reference_cases():
    from climt_amd.synthetic import make_columns, overcast
    from oracle.ref_driver import RefLW, RefSW
    from tools.pack_tables import read_blob
    from tools.synth_lw_tables import fill_reference_from_blob
    sw = RefSW()
    blob = read_blob(os.path.join(ROOT, "climt_amd", "data", "rrtmg_lw_data.bin"))
    lw = RefLW()
    lw.init(fill_tables=lambda r: fill_reference_from_blob(r, blob))
    base = dict(icld=1, iaer=0, adjes=1.0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1)
    cases = {}
    cases["clear_L60"] = (dict(ncol=48, nlay=60, cloudy=False, seed=11), {}, False)
    cases["clear_L30"] = (dict(ncol=32, nlay=30, cloudy=False, seed=12), {}, False)
    cases["overcast_L60"] = (dict(ncol=48, nlay=60, cloudy=True, seed=13), {"_overcast": True}, False)
    cases["mcica_kiss_random"] = (dict(ncol=48, nlay=60, cloudy=True, seed=14), dict(icld=1, irng=0, permuteseed=684), True)
    cases["mcica_kiss_maxrand"] = (dict(ncol=48, nlay=60, cloudy=True, seed=15), dict(icld=2, irng=0, permuteseed=112), True)
    cases["mcica_mt_max"] = (dict(ncol=24, nlay=40, cloudy=True, seed=16), dict(icld=3, irng=1, permuteseed=209652396), True)
    for name, (gen, extra, mcica) in cases.items():
        c = make_columns(**gen)
        if extra.pop("_overcast", False):
            c = overcast(c)
        c.update(base)
        c.update(extra)
        save = {"gen/" + k: np.array(v) for k, v in gen.items()}
        save.update({"flag/" + k: np.array(v) for k, v in c.items() if not isinstance(v, np.ndarray)})
        save["flag/_mcica"] = np.array(int(mcica))
        save["flag/_overcast"] = np.array(int(name.startswith("overcast")))
        r = sw.fluxes(c, mcica=mcica)
        for k in ("swuflx", "swdflx", "swhr", "swuflxc", "swdflxc", "swhrc"):
            save["sw/" + k] = r[k]
        # non-McICA LW: random overlap only (rtrn); McICA LW handles all overlaps
        cl = dict(c)
        if not mcica:
            cl["icld"] = 1
        r = lw.fluxes(cl, mcica=mcica)
        for k in ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc"):
            save["lw/" + k] = r[k]
        np.savez_compressed(os.path.join(OUT, "ref_%s.npz" % name), **save)
        print("reference case", name)


def lw_maxrand_columns(seed, ncol=40, nlay=60):
    """Columns for the non-McICA maximum/random overlap (rtrnmr) cases: every cloudy layer has its own fraction,
    some equal to a neighbour's, some layers overcast (the branches of the overlap-factor recursion)."""
    from climt_amd.synthetic import make_columns
    c = make_columns(ncol, nlay, cloudy=True, seed=seed)
    rng = np.random.default_rng(seed)
    f = rng.uniform(0.05, 1.0, c["cldfr"].shape)
    f[:, ::3] = np.round(f[:, ::3] * 4) / 4
    c["cldfr"] = np.where(c["cldfr"] > 0, np.clip(f, 0.01, 1.0), 0.0)
    c["cldfr"][:, :8] = np.where(c["cldfr"][:, :8] > 0, 1.0, 0.0)
    return c


def reference_rtrnmr_cases():
    """ref_lwmr_*.npz: longwave only (the non-McICA shortwave accepts clear/overcast layers only)."""
    from oracle.ref_driver import RefLW
    from tools.pack_tables import read_blob
    from tools.synth_lw_tables import fill_reference_from_blob
    blob = read_blob(os.path.join(ROOT, "climt_amd", "data", "rrtmg_lw_data.bin"))
    lw = RefLW()
    lw.init(fill_tables=lambda r: fill_reference_from_blob(r, blob))
    for name, seed, idrv, icld in (("maxrand", 31, 0, 2), ("maxrand_idrv", 32, 1, 2), ("maximum", 33, 0, 3)):
        c = lw_maxrand_columns(seed)
        c.update(icld=icld, iaer=0, inflg=2, iceflg=1, liqflg=1, idrv=idrv)
        r = lw.fluxes(c, mcica=False)
        save = {"flag/seed": np.array(seed), "flag/idrv": np.array(idrv), "flag/icld": np.array(icld), "in/cldfr": c["cldfr"]}
        for k, v in r.items():
            save["lw/" + k] = v
        np.savez_compressed(os.path.join(OUT, "ref_lwmr_%s.npz" % name), **save)
        print("rtrnmr case", name)


SOLVAR_CASES = [(isolvar, scon, ind, frac)
                for isolvar in (-1, 0, 1, 2, 3) for scon in (0.0, 1365.0)
                for ind, frac in (((1.0, 1.0), 0.2), ((1.2, 0.8), 0.01), ((1.2, 0.8), 0.2), ((1.2, 0.8), 0.7))
                if not (isolvar == 2 and scon > 0)]


def solvar_inputs(isolvar, scon, ind, frac):
    from climt_amd.synthetic import make_columns
    c = make_columns(12, 30, cloudy=False, seed=5)
    c.update(icld=0, iaer=0, dyofyr=30, scon=scon, isolvar=isolvar, inflg=0, iceflg=0, liqflg=0, adjes=1.0,
             solcycfrac=frac, indsolvar=np.array(ind), bndsolvar=np.linspace(0.9, 1.1, 14))
    if isolvar == 2:   # Mg / SB indices given directly
        c["indsolvar"] = np.array([0.16 * ind[0], 900.0 * ind[1]])
    return c


def reference_solvar_cases():
    """ref_sw_solvar.npz: every solar-variability method x internal/given solar constant x facular/sunspot amplitudes
    (amplitudes != 1 exercise the reference's per-column in-place rescaling) -> total-sky fluxes of 12 clear columns."""
    from oracle.ref_driver import RefSW
    sw = RefSW()
    sw.init()
    save = {}
    for i, case in enumerate(SOLVAR_CASES):
        r = sw.fluxes(solvar_inputs(*case), mcica=False)
        save["case%02d/swuflx" % i] = r["swuflx"]
        save["case%02d/swdflx" % i] = r["swdflx"]
    np.savez_compressed(os.path.join(OUT, "ref_sw_solvar.npz"), **save)
    print("solvar cases", len(SOLVAR_CASES))


def instellation_cases():
    """climt_cache_TestInstellation-{column,3d}.npz: the expected zenith angles of the reference's own golden caches
    (inputs are climt.get_grid's default latitude/longitude/time, regenerated by oracle/instellation_oracle.py)."""
    for desc in ("column", "3d"):
        v, dims, units = read_cache("TestInstellation-%s-0.cache" % desc)["zenith_angle"]
        np.savez_compressed(os.path.join(OUT, "climt_cache_TestInstellation-%s.npz" % desc), zenith_angle=v, dims=np.array(",".join(dims)), units=np.array(units))
        print("instellation cache", desc, v.shape)


def berger_cases():
    """climt_cache_TestBergerSolarInsolation-{column,3d}.npz: the expected outputs of the reference's golden caches."""
    for desc in ("column", "3d"):
        c = read_cache("TestBergerSolarInsolation-%s-0.cache" % desc)
        np.savez_compressed(os.path.join(OUT, "climt_cache_TestBergerSolarInsolation-%s.npz" % desc), **{k: v[0] for k, v in c.items()})
        print("berger cache", desc, {k: v[0].shape for k, v in c.items()})


def slab_surface_cases():
    """climt_cache_TestSlabSurface-{column,3d}.npz: default-state inputs (from the *_stepping-1 caches, which hold the full
    state) and the expected tendency / diagnostics of the reference's golden caches."""
    for desc in ("column", "3d"):
        state = read_cache("TestSlabSurface-%s_stepping-1.cache" % desc)
        save = {}
        for k, (v, dims, units) in state.items():
            if k == "time":
                continue
            if k == "area_type":      # char array (..., 100) -> strings
                v = np.array([b"".join(x).decode().strip() for x in v.reshape(-1, v.shape[-1])]).reshape(v.shape[:-1])
                dims = dims[:-1]
            save["state/%s/values" % k] = v
            save["state/%s/dims" % k] = np.array(",".join(dims))
            save["state/%s/units" % k] = np.array(units)
        for grp, idx in (("tend", 0), ("diag", 1)):
            for k, (v, dims, units) in read_cache("TestSlabSurface-%s-%d.cache" % (desc, idx)).items():
                save["%s/%s/values" % (grp, k)] = v
                save["%s/%s/dims" % (grp, k)] = np.array(",".join(dims))
                save["%s/%s/units" % (grp, k)] = np.array(units)
        np.savez_compressed(os.path.join(OUT, "climt_cache_TestSlabSurface-%s.npz" % desc), **save)
        print("slab surface cache", desc, len(save))



# ---- option coverage: aerosols, direct cloud optics, every ice / liquid parameterisation, grey surfaces -------------------
def _opt_base(seed, ncol=24, nlay=40, overcast_layers=False):
    from climt_amd.synthetic import make_columns, overcast
    c = make_columns(ncol, nlay, cloudy=True, seed=seed)
    if overcast_layers:
        c = overcast(c)
    c.pop("lat")
    c.update(icld=1, iaer=0, adjes=1.0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1, irng=0, permuteseed=37 + seed)
    return c


def _opt_aer6(c, rng):
    nlay, ncol = c["play"].shape
    # ECMWF aerosol optical thickness at 0.55 um, six types, concentrated in the lowest third of the column
    prof = np.exp(-np.arange(nlay) / (nlay / 6.0))[None, :, None]
    c["ecaer"] = np.ascontiguousarray(rng.uniform(0.0, 0.08, (6, nlay, ncol)) * prof)
    c["ecaer"][3, :, ::5] = 0.0          # a type absent in some columns
    c["ecaer"][:, nlay - 3:, :] = 0.0    # aerosol-free layers: the (0, 1, 0) branch of the mixing
    c["iaer"] = 6


def _opt_aer10(c, rng, nb):
    nlay, ncol = c["play"].shape
    prof = np.exp(-np.arange(nlay) / (nlay / 5.0))[None, :, None]
    c["tauaer"] = np.ascontiguousarray(rng.uniform(0.0, 0.05, (nb, nlay, ncol)) * prof)
    if nb == 14:
        c["ssaaer"] = np.ascontiguousarray(rng.uniform(0.75, 1.0, (nb, nlay, ncol)))
        c["asmaer"] = np.ascontiguousarray(rng.uniform(0.3, 0.8, (nb, nlay, ncol)))
        c["iaer"] = 10


def _opt_inflag0(c, rng, nb):
    nlay, ncol = c["play"].shape
    cld = c["cldfr"] > 0
    tau = rng.uniform(0.2, 6.0, (nlay, ncol, nb)) * cld[:, :, None]
    tau[:, ::4, :] *= 0.0                # cloud fraction > 0 with zero optical depth: the tauctot gate
    c["taucld"] = np.ascontiguousarray(tau)
    if nb == 14:
        g = rng.uniform(0.7, 0.9, (nlay, ncol, nb))
        c["ssacld"] = np.ascontiguousarray(rng.uniform(0.85, 0.999999, (nlay, ncol, nb)))
        c["asmcld"] = np.ascontiguousarray(g)
        c["fsfcld"] = np.ascontiguousarray(g * g)
    c["inflg"] = 0
    # the water paths stay in the state (cwp gate) but must not be used
    c["cicewp"] = c["cicewp"] * (rng.uniform(0, 1, (nlay, ncol)) > 0.5)
    c["cicewp"][:, ::8] = 0.0; c["cliqwp"][:, ::8] = 0.0     # neither water path nor optical depth: gate closed


def _opt_sizes(c, rng, iceflg, liqflg):
    nlay, ncol = c["play"].shape
    lo, hi = {0: (10.0, 120.0), 1: (13.0, 130.0), 2: (5.0, 131.0), 3: (5.0, 140.0)}[iceflg]
    c["reice"] = np.ascontiguousarray(rng.uniform(lo, hi, (nlay, ncol)))
    c["reice"][:, 0] = lo; c["reice"][:, 1] = hi          # the table end points (index clamps)
    c["reliq"] = np.ascontiguousarray(rng.uniform(2.5, 60.0, (nlay, ncol)))
    c["reliq"][:, 2] = 2.5; c["reliq"][:, 3] = 60.0
    c["iceflg"], c["liqflg"] = iceflg, liqflg


def _opt_emis(c, rng):
    ncol = c["play"].shape[1]
    c["emis"] = np.ascontiguousarray(rng.uniform(0.90, 0.99, (16, ncol)))


def _opt_abundant(c, rng):
    """CO2 and N2O well above the reference profiles in most columns: taumol's "too abundant" column adjustments (bands 3, 6, 7,
    8, 9, 13: adjfac = a + (rat - a)**e once rat exceeds 1.5 / 3.0, rrtmg_lw_taumol.f90:704-720 etc.) take their pow() branch
    in some layers and columns and not in others (the first columns keep the defaults)."""
    nlay, ncol = c["play"].shape
    fco2 = np.concatenate([[1.0, 1.0], rng.uniform(1.0, 8.0, ncol - 2)])
    fn2o = np.concatenate([[1.0, 1.0], rng.uniform(1.0, 5.0, ncol - 2)])
    c["co2"] = np.ascontiguousarray(c["co2"] * fco2[None, :] * rng.uniform(0.8, 1.2, (nlay, ncol)))
    c["n2o"] = np.ascontiguousarray(c["n2o"] * fn2o[None, :] * rng.uniform(0.8, 1.2, (nlay, ncol)))
    _opt_emis(c, rng)


def option_cases():
    """name -> (spectrum, mcica, inputs): every non-default option of the path with its own seeded inputs."""
    cases = {}
    def add(name, spectrum, mcica, seed, build, overcast_layers=False, **flags):
        rng = np.random.default_rng(9000 + seed)
        c = _opt_base(seed, overcast_layers=overcast_layers)
        build(c, rng)
        c.update(flags)
        cases[name] = (spectrum, mcica, c)
    # shortwave (reference k-distribution data: physical parity)
    add("sw_aer6_clear", "sw", False, 1, _opt_aer6, icld=0)
    add("sw_aer6_mcica", "sw", True, 2, _opt_aer6, icld=2)
    add("sw_aer10_overcast", "sw", False, 3, lambda c, r: _opt_aer10(c, r, 14), overcast_layers=True)
    add("sw_aer10_mcica", "sw", True, 4, lambda c, r: _opt_aer10(c, r, 14), icld=1)
    add("sw_inflag0_overcast", "sw", False, 5, lambda c, r: _opt_inflag0(c, r, 14), overcast_layers=True)
    add("sw_inflag0_mcica", "sw", True, 6, lambda c, r: _opt_inflag0(c, r, 14), icld=2)
    for ice in (2, 3):
        add("sw_ice%d_overcast" % ice, "sw", False, 10 + ice, lambda c, r, ice=ice: _opt_sizes(c, r, ice, 1), overcast_layers=True)
        add("sw_ice%d_mcica" % ice, "sw", True, 20 + ice, lambda c, r, ice=ice: _opt_sizes(c, r, ice, 1), icld=3 if ice == 3 else 1)
    # longwave (synthetic k-tables: algorithm parity)
    def emis_aer(c, r):
        _opt_emis(c, r); _opt_aer10(c, r, 16)
    add("lw_emis_aer_clear", "lw", False, 31, emis_aer, icld=0)
    add("lw_emis_aer_random", "lw", False, 32, emis_aer, icld=1)
    add("lw_emis_aer_maxrand_idrv", "lw", False, 33, emis_aer, icld=2, idrv=1)
    add("lw_emis_aer_mcica", "lw", True, 34, emis_aer, icld=2, idrv=1)
    add("lw_inflag0_random", "lw", False, 35, lambda c, r: (_opt_inflag0(c, r, 16), _opt_emis(c, r)), icld=1)
    add("lw_inflag0_maxrand", "lw", False, 36, lambda c, r: _opt_inflag0(c, r, 16), icld=2)
    add("lw_inflag0_mcica", "lw", True, 37, lambda c, r: (_opt_inflag0(c, r, 16), _opt_emis(c, r)), icld=1)
    add("lw_inflag1_random", "lw", False, 38, lambda c, r: _opt_emis(c, r), icld=1, inflg=1)
    add("lw_abundant_clear", "lw", False, 81, _opt_abundant, icld=0)
    add("lw_abundant_mcica", "lw", True, 82, _opt_abundant, icld=2)
    for ice, liq in ((0, 0), (0, 1), (1, 0), (2, 1), (3, 1), (2, 0), (3, 0)):
        add("lw_ice%d_liq%d_random" % (ice, liq), "lw", False, 40 + 4 * ice + liq, lambda c, r, i=ice, q=liq: (_opt_sizes(c, r, i, q), _opt_emis(c, r)), icld=1)
        add("lw_ice%d_liq%d_mcica" % (ice, liq), "lw", True, 60 + 4 * ice + liq, lambda c, r, i=ice, q=liq: _opt_sizes(c, r, i, q), icld=2)
    return cases


def reference_option_cases(only=None):
    """ref_opt_<name>.npz: inputs stored IN the fixture (self-contained), outputs of the reference Fortran."""
    from oracle.ref_driver import RefLW, RefSW
    from tools.pack_tables import read_blob
    from tools.synth_lw_tables import fill_reference_from_blob
    sw = RefSW()
    blob = read_blob(os.path.join(ROOT, "climt_amd", "data", "rrtmg_lw_data.bin"))
    lw = RefLW()
    lw.init(fill_tables=lambda r: fill_reference_from_blob(r, blob))
    for name, (spectrum, mcica, c) in option_cases().items():
        if only and not any(o in name for o in only):
            continue
        save = {"in/" + k: v for k, v in c.items() if isinstance(v, np.ndarray)}
        save.update({"flag/" + k: np.array(v) for k, v in c.items() if not isinstance(v, np.ndarray)})
        save["flag/_mcica"] = np.array(int(mcica))
        r = (sw if spectrum == "sw" else lw).fluxes(dict(c), mcica=mcica)
        keys = ("swuflx", "swdflx", "swhr", "swuflxc", "swdflxc", "swhrc") if spectrum == "sw" else \
               ("uflx", "dflx", "hr", "uflxc", "dflxc", "hrc") + (("duflx_dt", "duflxc_dt") if c.get("idrv") else ())
        for k in keys:
            save["%s/%s" % (spectrum, k)] = r[k]
        assert all(np.all(np.isfinite(r[k])) for k in keys), name
        np.savez_compressed(os.path.join(OUT, "ref_opt_%s.npz" % name), **save)
        print("option case", name, {k: float(np.abs(r[k]).max()) for k in keys[:2]})


def cache_outputs_only(cls, desc):
    """climt_cacheout_<Class>-<desc>.npz: the expected outputs of a reference cache whose input state is a missing blob
    (`*_stepping-1.cache` absent: .MISSING_LARGE_BLOBS) but is the plain default state of the reference's test
    (tests/test_components.py:250-255: get_default_state([component], get_grid(nx=32, ny=16, nz=28))), which
    climt_amd.get_default_state reproduces (pinned by the other caches).  Tendencies, diagnostics, and the diagnostics of the
    10 s Adams-Bashforth step (`*_stepping-0.cache`)."""
    save = {}
    for grp, fn in (("tend", "%s-%s-0.cache"), ("diag", "%s-%s-1.cache"), ("stepdiag", "%s-%s_stepping-0.cache")):
        path = fn % (cls, desc)
        if not os.path.exists(os.path.join(CACHE, path)):
            continue
        for k, (v, d, u) in read_cache(path).items():
            save["%s/%s/values" % (grp, k)] = v
            save["%s/%s/dims" % (grp, k)] = np.array(",".join(d))
            save["%s/%s/units" % (grp, k)] = np.array(u)
    np.savez_compressed(os.path.join(OUT, "climt_cacheout_%s-%s.npz" % (cls, desc)), **save)
    print("cache outputs", cls, desc, len(save))


# ---- the longwave drop-in CLASS: values a user of RRTMGLongwave()(state) gets ----------------------------------------------
# An independent numpy restatement of what happens between a sympl state and the reference binder's argument list
# (sympl's extraction: units + dim order, climt/_components/rrtmg/lw/component.py:374-447, climt/_core/util.py:47-142,
# _rrtmg_lw.pyx:137-282), feeding the reference Fortran (oracle/_ref; SYNTHETIC k-tables while rrtmg_lw_k_g.f90 is a missing
# blob).  Nothing of climt_amd's host layer (_sympl_compat, rrtmg/longwave.py, _util.py) is used here: the fixtures pin it.
_LW_CLASS_INPUTS = {   # name -> (target dims, factor from the cached state's units to the component's units)
    "air_pressure": (("mid_levels", "*"), {"Pa": 0.01, "mbar": 1.0}),
    "air_pressure_on_interface_levels": (("interface_levels", "*"), {"Pa": 0.01, "mbar": 1.0}),
    "air_temperature": (("mid_levels", "*"), {"degK": 1.0}),
    "air_temperature_on_interface_levels": (("interface_levels", "*"), {"degK": 1.0}),
    "surface_temperature": (("*",), {"degK": 1.0}),
    "specific_humidity": (("mid_levels", "*"), {"dimensionless": 1.0, "g/g": 1.0, "kg/kg": 1.0}),
    "surface_longwave_emissivity": (("num_longwave_bands", "*"), {"dimensionless": 1.0}),
    "cloud_area_fraction_in_atmosphere_layer": (("mid_levels", "*"), {"dimensionless": 1.0}),
    "longwave_optical_thickness_due_to_cloud": (("mid_levels", "*", "num_longwave_bands"), {"dimensionless": 1.0}),
    "mass_content_of_cloud_ice_in_atmosphere_layer": (("mid_levels", "*"), {"kg/m**2": 1000.0, "kg m^-2": 1000.0, "g m^-2": 1.0}),
    "mass_content_of_cloud_liquid_water_in_atmosphere_layer": (("mid_levels", "*"), {"kg/m**2": 1000.0, "kg m^-2": 1000.0, "g m^-2": 1.0}),
    "cloud_ice_particle_size": (("mid_levels", "*"), {"µm": 1.0, "micrometer": 1.0}),
    "cloud_water_droplet_radius": (("mid_levels", "*"), {"µm": 1.0, "micrometer": 1.0}),
    "longwave_optical_thickness_due_to_aerosol": (("num_longwave_bands", "mid_levels", "*"), {"dimensionless": 1.0}),
}
for _gas in ("ozone", "carbon_dioxide", "methane", "nitrous_oxide", "oxygen", "cfc11", "cfc12", "cfc22", "carbon_tetrachloride"):
    _LW_CLASS_INPUTS["mole_fraction_of_%s_in_air" % _gas] = (("mid_levels", "*"), {"dimensionless": 1.0})
_LW_OVERLAP = {"clear_only": 0, "random": 1, "maximum_random": 2, "maximum": 3}           # rrtmg_common.py:8-13
_LW_CLOUD_PROPS = {"direct_input": 0, "single_cloud_type": 1, "liquid_and_ice_clouds": 2}     # :20-24
_LW_ICE = {"ebert_curry_one": 0, "ebert_curry_two": 1, "key_streamer_manual": 2, "fu": 3}    # :31-36
_LW_LIQ = {"radius_independent_absorption": 0, "radius_dependent_absorption": 1}             # :43-46


def lw_class_reference(state, kwargs, lw):
    """state: {name: (values, dims, units)}; kwargs: RRTMGLongwave's constructor arguments -> ({group: {name: (values, dims,
    units)}}) as the reference class would return it, computed by the reference Fortran `lw` (an initialised RefLW)."""
    named = ("mid_levels", "interface_levels", "num_longwave_bands")
    wild = [d for d in state["air_pressure"][1] if d not in named]            # sympl: wildcard dims in order of first appearance
    wshape = tuple(state["air_pressure"][0].shape[state["air_pressure"][1].index(d)] for d in wild)
    raw = {}
    for name, (target, factors) in _LW_CLASS_INPUTS.items():
        if name not in state:
            continue
        v, dims, units = state[name]
        order = []
        for t in target:
            order += [dims.index(d) for d in wild] if t == "*" else [dims.index(t)]
        a = np.transpose(np.asarray(v, dtype=np.float64), order)
        # (the named dims keep their lengths; the wildcard dims collapse, C order, into the column index)
        shp, i = [], 0
        for t in target:
            if t == "*":
                shp.append(int(np.prod(wshape))); i += len(wild)
            else:
                shp.append(a.shape[i]); i += 1
        raw[name] = np.ascontiguousarray(a.reshape(shp) * factors[units])
    p, pi, t = raw["air_pressure"], raw["air_pressure_on_interface_levels"], raw["air_temperature"]
    nlay, ncol = t.shape
    if kwargs.get("calculate_interface_temperature", True):
        # climt/_core/util.py:89-142, restated
        tint = np.zeros((nlay + 1, ncol))
        lp = np.log(p)
        w = (np.log(pi[1:-1]) - lp[1:]) / (lp[:-1] - lp[1:])
        tint[1:-1] = t[1:] - w * (t[1:] - t[:-1])
        tint[0] = raw["surface_temperature"]
        tint[-1] = t[-1]
    else:
        tint = raw["air_temperature_on_interface_levels"]
    mcica = bool(kwargs.get("mcica", False))
    overlap = kwargs.get("cloud_overlap_method") or "random"                          # lw/component.py:296-297
    c = dict(play=p, plev=pi, tlay=t, tlev=tint, tsfc=raw["surface_temperature"],
             h2o=raw["specific_humidity"] * 28.964 / 18.02,                           # util.py:86
             o3=raw["mole_fraction_of_ozone_in_air"], co2=raw["mole_fraction_of_carbon_dioxide_in_air"],
             ch4=raw["mole_fraction_of_methane_in_air"], n2o=raw["mole_fraction_of_nitrous_oxide_in_air"],
             o2=raw["mole_fraction_of_oxygen_in_air"], cfc11=raw["mole_fraction_of_cfc11_in_air"],
             cfc12=raw["mole_fraction_of_cfc12_in_air"], cfc22=raw["mole_fraction_of_cfc22_in_air"],
             ccl4=raw["mole_fraction_of_carbon_tetrachloride_in_air"], emis=raw["surface_longwave_emissivity"],
             cldfr=raw["cloud_area_fraction_in_atmosphere_layer"], taucld=raw["longwave_optical_thickness_due_to_cloud"],
             cicewp=raw["mass_content_of_cloud_ice_in_atmosphere_layer"],
             cliqwp=raw["mass_content_of_cloud_liquid_water_in_atmosphere_layer"],
             reice=raw["cloud_ice_particle_size"], reliq=raw["cloud_water_droplet_radius"],
             tauaer=raw["longwave_optical_thickness_due_to_aerosol"],
             icld=_LW_OVERLAP[overlap.lower()], idrv=0,
             inflg=_LW_CLOUD_PROPS[kwargs.get("cloud_optical_properties", "liquid_and_ice_clouds").lower()],
             iceflg=_LW_ICE[kwargs.get("cloud_ice_properties", "ebert_curry_two").lower()],
             liqflg=_LW_LIQ[kwargs.get("cloud_liquid_water_properties", "radius_dependent_absorption").lower()])
    if mcica:
        # the seed drawn per call (lw/component.py:415-424) after the reference tests' np.random.seed(0) (test_components.py:148)
        np.random.seed(0)
        irng = {"kissvec": 0, "mersenne_twister": 1}[kwargs.get("random_number_generator", "mersenne_twister").lower()]
        c.update(irng=irng, permuteseed=int(np.random.randint(0, 1024) if irng == 0 else np.random.randint(0, 2 ** 31 - 1)))
    r = lw.fluxes(c, mcica=mcica)
    il, ml = ("interface_levels",) + tuple(wild), ("mid_levels",) + tuple(wild)
    back = lambda a: np.ascontiguousarray(a.reshape((a.shape[0],) + wshape))
    diag = {"upwelling_longwave_flux_in_air": (back(r["uflx"]), il, "W m^-2"),
            "downwelling_longwave_flux_in_air": (back(r["dflx"]), il, "W m^-2"),
            "upwelling_longwave_flux_in_air_assuming_clear_sky": (back(r["uflxc"]), il, "W m^-2"),
            "downwelling_longwave_flux_in_air_assuming_clear_sky": (back(r["dflxc"]), il, "W m^-2"),
            "air_temperature_tendency_from_longwave_assuming_clear_sky": (back(r["hrc"]), ml, "degK day^-1"),
            "air_temperature_tendency_from_longwave": (back(r["hr"]), ml, "degK day^-1")}      # alias, lw/component.py:518-520
    return {"tend": {"air_temperature": (back(r["hr"]), ml, "degK day^-1")}, "diag": diag}, c


def _lw_class_perturbed_states():
    """States that a wrong axis, a wrong unit factor or a wrong flag cannot survive: every optional longwave input is
    non-trivial and differs along every axis (the reference's default states have emissivity 1, no aerosol, no cloud optical
    depth, q = 0).  Built on the grid of the TestRRTMGLongwaveMCICA 3-d cache state (5 x 10 x 28)."""
    import json
    z = np.load(os.path.join(OUT, "climt_cache_TestRRTMGLongwaveMCICA-3d.npz"))
    base = {k.split("/")[1]: (z[k], tuple(str(z[k[:-6] + "dims"]).split(",")), str(z[k[:-6] + "units"]))
            for k in z.files if k.startswith("state/") and k.endswith("/values")}
    rng = np.random.default_rng(20260928)
    L, ny, nx = base["air_temperature"][0].shape
    def put(st, name, v):
        st[name] = (np.ascontiguousarray(v), st[name][1], st[name][2])
    def common(st):
        p = st["air_pressure"][0]
        t = np.maximum(288.0 * (p / p[0]) ** 0.19 + rng.uniform(-3, 3, p.shape), 205.0)
        put(st, "air_temperature", t)
        put(st, "surface_temperature", t[0] + rng.uniform(-2.0, 6.0, (ny, nx)))
        put(st, "specific_humidity", np.maximum(3e-6, 0.012 * (p / p[0]) ** 3 * rng.uniform(0.5, 1.0, p.shape)))
        put(st, "mole_fraction_of_methane_in_air", np.full(p.shape, 1.7e-6) * rng.uniform(0.9, 1.1, p.shape))
        put(st, "mole_fraction_of_nitrous_oxide_in_air", np.full(p.shape, 3.0e-7) * rng.uniform(0.9, 1.1, p.shape))
        for gas, ppb in (("cfc11", 0.25), ("cfc12", 0.5), ("cfc22", 0.1), ("carbon_tetrachloride", 0.1)):
            put(st, "mole_fraction_of_%s_in_air" % gas, np.full(p.shape, ppb * 1e-9) * rng.uniform(0.8, 1.2, p.shape))
        put(st, "surface_longwave_emissivity", rng.uniform(0.85, 1.0, (16, ny, nx)))
        prof = np.exp(-np.arange(L) / 5.0)[None, :, None, None]
        put(st, "longwave_optical_thickness_due_to_aerosol", rng.uniform(0.0, 0.06, (16, L, ny, nx)) * prof)
        f = np.zeros((L, ny, nx))
        f[6:12] = rng.choice([0.0, 0.25, 0.5, 1.0], (6, ny, nx))
        f[15:19] = rng.uniform(0.05, 0.9, (4, ny, nx)) * (rng.uniform(0, 1, (4, ny, nx)) > 0.3)
        put(st, "cloud_area_fraction_in_atmosphere_layer", f)
        put(st, "mass_content_of_cloud_liquid_water_in_atmosphere_layer", np.where((f > 0) & (t > 253.0), rng.uniform(0.02, 0.08, f.shape), 0.0))
        put(st, "mass_content_of_cloud_ice_in_atmosphere_layer", np.where((f > 0) & (t < 263.0), rng.uniform(0.005, 0.03, f.shape), 0.0))
        put(st, "cloud_ice_particle_size", rng.uniform(15.0, 120.0, f.shape))
        put(st, "cloud_water_droplet_radius", rng.uniform(4.0, 40.0, f.shape))
        put(st, "longwave_optical_thickness_due_to_cloud", rng.uniform(0.2, 5.0, (L, ny, nx, 16)) * (f > 0)[..., None])
        return st
    cases = {}
    cases["direct_input_random"] = (common(dict(base)), dict(cloud_optical_properties="direct_input", cloud_overlap_method="random"))
    cases["single_cloud_type_maxrand"] = (common(dict(base)), dict(cloud_optical_properties="single_cloud_type", cloud_overlap_method="maximum_random"))
    cases["liquid_ice_fu_maximum"] = (common(dict(base)), dict(cloud_ice_properties="fu", cloud_overlap_method="maximum"))
    cases["mcica_kissvec_maxrand_streamer"] = (common(dict(base)), dict(mcica=True, random_number_generator="kissvec", cloud_overlap_method="maximum_random",
                                                                       cloud_ice_properties="key_streamer_manual"))
    cases["mcica_twister_direct_input"] = (common(dict(base)), dict(mcica=True, cloud_optical_properties="direct_input"))
    # external interface temperatures + every array handed over with its axes in ANOTHER order (lon, lat, levels[, bands])
    st = common(dict(base))
    tint = np.concatenate([st["surface_temperature"][0][None], 0.5 * (st["air_temperature"][0][1:] + st["air_temperature"][0][:-1]) + rng.uniform(-1, 1, (L - 1, ny, nx)),
                           st["air_temperature"][0][-1:]], axis=0)
    st["air_temperature_on_interface_levels"] = (tint, ("interface_levels", "lat", "lon"), "degK")
    for k, (v, dims, units) in list(st.items()):
        if "lat" in dims and "lon" in dims:
            new = tuple(sorted(dims, key=lambda d: {"lon": 0, "lat": 1}.get(d, 2 + dims.index(d))))
            st[k] = (np.ascontiguousarray(np.transpose(v, [dims.index(d) for d in new])), new, units)
    cases["external_tint_axes_reordered"] = (st, dict(calculate_interface_temperature=False, cloud_overlap_method="random"))
    return cases


def reference_lw_class_cases():
    """ref_lwclass_<name>.npz: what climt.RRTMGLongwave(**kwargs)(state) returns when its Fortran runs on this build's
    (synthetic) longwave tables -- (i) the four states of the reference's own longwave cache classes
    (tests/test_components.py:435-480), (ii) the perturbed states above, stored in the file."""
    import json
    from oracle.ref_driver import RefLW
    from tools.pack_tables import read_blob
    from tools.synth_lw_tables import fill_reference_from_blob
    blob = read_blob(os.path.join(ROOT, "climt_amd", "data", "rrtmg_lw_data.bin"))
    lw = RefLW()
    lw.init(fill_tables=lambda r: fill_reference_from_blob(r, blob))
    todo = {}
    for cls, desc, kw in (("TestRRTMGLongwave", "column", {}),
                          ("TestRRTMGLongwaveWithClouds", "column", dict(cloud_optical_properties="single_cloud_type")),
                          ("TestRRTMGLongwaveWithExternalInterfaceTemperature", "column", dict(calculate_interface_temperature=False)),
                          ("TestRRTMGLongwaveMCICA", "3d", dict(mcica=True))):
        z = np.load(os.path.join(OUT, "climt_cache_%s-%s.npz" % (cls, desc)))
        st = {k.split("/")[1]: (z[k], tuple(str(z[k[:-6] + "dims"]).split(",")), str(z[k[:-6] + "units"]))
              for k in z.files if k.startswith("state/") and k.endswith("/values")}
        todo["%s-%s" % (cls, desc)] = (st, kw, False)
    for name, (st, kw) in _lw_class_perturbed_states().items():
        todo[name] = (st, kw, True)
    for name, (st, kw, store_state) in todo.items():
        exp, c = lw_class_reference(st, kw, lw)
        save = {"kwargs": np.array(json.dumps(kw)), "synthetic_tables": np.array(int(np.ravel(blob.get("lw/meta/synthetic", np.array([0])))[0]))}
        if store_state:
            for k, (v, d, u) in st.items():
                save["state/%s/values" % k] = v
                save["state/%s/dims" % k] = np.array(",".join(d))
                save["state/%s/units" % k] = np.array(u)
        for grp, dd in exp.items():
            for k, (v, d, u) in dd.items():
                assert np.all(np.isfinite(v)), (name, k)
                save["%s/%s/values" % (grp, k)] = v
                save["%s/%s/dims" % (grp, k)] = np.array(",".join(d))
                save["%s/%s/units" % (grp, k)] = np.array(u)
        np.savez_compressed(os.path.join(OUT, "ref_lwclass_%s.npz" % name), **save)
        print("longwave class case", name, kw, "OLR", float(exp["diag"]["upwelling_longwave_flux_in_air"][0][-1].mean()),
              "cloud effect", float(np.abs(exp["diag"]["upwelling_longwave_flux_in_air"][0] - exp["diag"]["upwelling_longwave_flux_in_air_assuming_clear_sky"][0]).max()))


def pin_input_hashes():
    """tests/golden/input_hashes.json: sha256 of the inputs the generator (climt_amd.synthetic) produces for every fixture that
    stores generator arguments instead of inputs -- re-pinned whenever the fixtures are regenerated (tests/helpers.py checks)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    fn = os.path.join(OUT, "input_hashes.json")
    names, got = json.load(open(fn)), {}
    helpers._check_pinned_inputs = lambda name, c: got.__setitem__(name, helpers.input_hash(c))
    for n in sorted(names):
        if n.startswith("ref_lwmr_"):
            helpers.load_lwmr_case(n[len("ref_lwmr_"):])
        else:
            helpers.load_ref_case(n[len("ref_"):])
    assert set(got) == set(names)
    json.dump(got, open(fn, "w"), indent=1, sort_keys=True)
    print("input hashes pinned:", sum(got[k] != names[k] for k in got), "changed of", len(got))


if __name__ == "__main__":
    if sys.argv[1:] == ["lwclass"]:
        reference_lw_class_cases()
        sys.exit(0)
    if sys.argv[1:2] == ["options"]:      # options <substring> ...: only those option fixtures
        reference_option_cases(only=sys.argv[2:])
        sys.exit(0)
    n = 0
    for cls in ("TestRRTMGLongwave", "TestRRTMGLongwaveMCICA", "TestRRTMGLongwaveWithClouds",
                "TestRRTMGLongwaveWithExternalInterfaceTemperature", "TestRRTMGShortwave", "TestRRTMGShortwaveMCICA"):
        for desc in ("column", "3d"):
            r = cache_case(cls, desc)
            print(cls, desc, "->", r)
    reference_cases()
    reference_rtrnmr_cases()
    reference_solvar_cases()
    instellation_cases()
    berger_cases()
    slab_surface_cases()
    reference_option_cases()
    for cls in ("TestRRTMGShortwave", "TestRRTMGLongwave"):
        cache_outputs_only(cls, "3d")
    reference_lw_class_cases()
    pin_input_hashes()
