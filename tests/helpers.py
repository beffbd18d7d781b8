"""Shared test helpers: golden-fixture loaders, the host emulator of the device functions (CPU tests),
and state builders.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import datetime
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from climt_amd._lib import LwArgs, SwArgs, LW_OUT, SW_OUT, SW_DATA, LW_DATA, _LW_FIELDS, _LW_FLAGS, _SW_FIELDS, _SW_FLAGS  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_SO = os.path.join(ROOT, "tests", "_emu", "librrtmg_emu.so")

CONSTANTS = dict(pi=np.pi, grav=9.80665, planck=6.62607004e-27, boltz=1.38064852e-16, clight=2.99792458e10,
                 avogad=6.022140857e23, alosmt=2.6867774e19, gascon=8.3144598e7, sbcnst=5.670367e-12, secdy=86400.0)
CPDAIR = 1004.64
_CONST_VEC = np.array([CONSTANTS[n] for n in "pi grav planck boltz clight avogad alosmt gascon sbcnst secdy".split()])

_emu = None


def emu_lib():
    """Host emulation of the device functions (tests/emu); built on demand with hipcc (no GPU needed)."""
    global _emu
    if _emu is None:
        srcs = [os.path.join(ROOT, "tests", "emu", f) for f in ("emu_sw.hip", "emu_lw.hip")]
        srcs += [os.path.join(ROOT, "climt_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "climt_amd", "csrc"))]
        if not os.path.exists(EMU_SO) or os.path.getmtime(EMU_SO) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build.sh")])
        _emu = C.CDLL(EMU_SO)
        _emu.emu_get_table.restype = C.c_long
        _emu.emu_get_table.argtypes = [C.c_char_p, C.c_char_p, C.c_double, C.c_char_p, C.c_void_p, C.c_long]
    return _emu


def _fill(a, inp, fields, flags, keep):
    for k, f in flags.items():
        if k in inp:
            setattr(a, f, int(inp[k]))
    # the unit factors the library applies on the device (rrtmg_host_inputs.h) are applied here with numpy -- the same
    # operations, one rounding each
    scale = {"play": ("pressure_scale",), "plev": ("pressure_scale",), "cicewp": ("water_path_scale",), "cliqwp": ("water_path_scale",),
             "h2o": ("h2o_mul", "h2o_div")}
    for k, f in fields.items():
        v = inp.get(k)
        if v is None:
            continue
        arr = np.ascontiguousarray(v, dtype=np.float64)
        names = scale.get(k)
        if names and inp.get(names[0]):
            arr = arr * float(inp[names[0]])
            if len(names) > 1 and inp.get(names[1]):
                arr = arr / float(inp[names[1]])
        keep.append(arr)
        setattr(a, f, arr.ctypes.data)


class EmuContext:
    """Drop-in for climt_amd._lib.Context that runs the device functions on the host (tests only)."""

    def __init__(self, device=0):
        self.lib = emu_lib()
        self.device = device

    def set_constants(self, **k):
        pass

    def sw_init(self, cpdair, blob=None):
        self.cpd_sw = cpdair
        self.sw_blob = blob or SW_DATA

    def lw_init(self, cpdair, blob=None):
        self.cpd_lw = cpdair
        self.lw_blob = blob or os.environ.get("RRTMG_HIP_LW_DATA") or LW_DATA      # as climt_amd._lib.Context.lw_init

    def lw_tables_synthetic(self):
        from tools.pack_tables import read_blob      # the flag the product reads at init (rrtmg_tables.cpp: "lw/meta/synthetic")
        return bool(int(np.ravel(read_blob(getattr(self, "lw_blob", LW_DATA)).get("lw/meta/synthetic", np.array([0])))[0]))

    def close(self):
        pass

    def sw_fluxes(self, inp, mcica=False, out=None, memspace=0):
        nlay, ncol = inp["play"].shape
        a, keep = SwArgs(), []
        a.ncol, a.nlay, a.memspace, a.mcica = ncol, nlay, 0, int(bool(mcica))
        a.icld, a.inflgsw, a.iceflgsw, a.liqflgsw, a.dyofyr = 1, 2, 1, 1, 1
        a.adjes, a.scon, a.solcycfrac = float(inp.get("adjes", 1.0)), float(inp.get("scon", 1367.0)), float(inp.get("solcycfrac", 0.0))
        _fill(a, inp, _SW_FIELDS, _SW_FLAGS, keep)
        if out is None:
            out = {k: np.zeros((nlay + lev, ncol)) for k, lev in SW_OUT}
        for k, _ in SW_OUT:
            setattr(a, k, out[k].ctypes.data)
        eb = C.create_string_buffer(512)
        rc = self.lib.emu_sw_fluxes(C.byref(a), getattr(self, "sw_blob", SW_DATA).encode(), C.c_double(CPDAIR), _CONST_VEC.ctypes.data_as(C.c_void_p), eb, 512)
        if rc:
            from climt_amd._lib import RRTMGError
            raise RRTMGError(rc, eb.value.decode())
        return out

    def lw_fluxes(self, inp, mcica=False, out=None, memspace=0):
        nlay, ncol = inp["play"].shape
        a, keep = LwArgs(), []
        a.ncol, a.nlay, a.memspace, a.mcica = ncol, nlay, 0, int(bool(mcica))
        a.icld, a.inflglw, a.iceflglw, a.liqflglw = 1, 2, 1, 1
        _fill(a, inp, _LW_FIELDS, _LW_FLAGS, keep)
        if out is None:
            out = {k: np.zeros((nlay + lev, ncol)) for k, lev in LW_OUT}
            if a.idrv:
                out["duflx_dt"] = np.zeros((nlay + 1, ncol))
                out["duflxc_dt"] = np.zeros((nlay + 1, ncol))
        for k in out:
            setattr(a, k, out[k].ctypes.data)
        eb = C.create_string_buffer(512)
        rc = self.lib.emu_lw_fluxes(C.byref(a), getattr(self, "lw_blob", LW_DATA).encode(), C.c_double(CPDAIR), _CONST_VEC.ctypes.data_as(C.c_void_p), eb, 512)
        if rc:
            from climt_amd._lib import RRTMGError
            raise RRTMGError(rc, eb.value.decode())
        return out

    def mcica_mask(self, which, play, cldfrac, icld, permuteseed, irng):
        nlay, ncol = play.shape
        nsub = 112 if which == "sw" else 140
        out = np.zeros((nlay, ncol, nsub))
        p, c = np.ascontiguousarray(play, dtype=np.float64), np.ascontiguousarray(cldfrac, dtype=np.float64)
        self.lib.emu_mask(0 if which == "sw" else 1, ncol, nlay, int(icld), int(permuteseed), int(irng),
                          p.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def get_table(self, name):
        which = name.split("/")[0]
        blob = (SW_DATA if which == "sw" else LW_DATA).encode()
        n = self.lib.emu_get_table(which.encode(), blob, CPDAIR, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n)
        self.lib.emu_get_table(which.encode(), blob, CPDAIR, name.encode(), out.ctypes.data, n)
        return out


# ---- golden fixtures -------------------------------------------------------------------------
REF_CASES = ("clear_L60", "clear_L30", "overcast_L60", "mcica_kiss_random", "mcica_kiss_maxrand", "mcica_mt_max")


def input_hash(c):
    """sha256 over every input array (name, shape, bytes) and scalar of a boundary-level input dict."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(c):
        v = c[k]
        if isinstance(v, np.ndarray):
            a = np.ascontiguousarray(v, dtype=np.float64)
            h.update(("%s%s" % (k, a.shape)).encode()); h.update(a.tobytes())
        else:
            h.update(("%s=%r" % (k, float(v))).encode())
    return h.hexdigest()


def _check_pinned_inputs(name, c):
    """The older fixtures store generator arguments, not inputs: tests/golden/input_hashes.json pins what the generator
    (climt_amd.synthetic) must reproduce, so that a change there cannot silently redefine golden inputs."""
    import json
    want = json.load(open(os.path.join(GOLDEN, "input_hashes.json")))[name]
    got = input_hash(c)
    assert got == want, "inputs of fixture %s changed (climt_amd.synthetic no longer reproduces them): %s != %s" % (name, got, want)


def load_ref_case(name):
    """-> (inputs dict at the C-ABI boundary, mcica flag, expected {'sw': {...}, 'lw': {...}})."""
    from climt_amd.synthetic import make_columns, overcast
    z = np.load(os.path.join(GOLDEN, "ref_%s.npz" % name))
    gen = {k[4:]: z[k].item() for k in z.files if k.startswith("gen/")}
    gen["cloudy"] = bool(gen["cloudy"])
    c = make_columns(**gen)
    flags = {k[5:]: z[k].item() for k in z.files if k.startswith("flag/")}
    if flags.pop("_overcast"):
        c = overcast(c)
    mcica = bool(flags.pop("_mcica"))
    c.update(flags)
    _check_pinned_inputs("ref_" + name, c)
    exp = {"sw": {k[3:]: z[k] for k in z.files if k.startswith("sw/")}, "lw": {k[3:]: z[k] for k in z.files if k.startswith("lw/")}}
    return c, mcica, exp


def load_cache_case(cls, desc):
    """Reference golden cache -> (state of DataArrays, expected tendencies, expected diagnostics)."""
    from climt_amd._sympl_compat import DataArray
    z = np.load(os.path.join(GOLDEN, "climt_cache_%s-%s.npz" % (cls, desc)))
    groups = {"state": {}, "tend": {}, "diag": {}}
    for k in z.files:
        grp, name, what = k.split("/")
        groups[grp].setdefault(name, {})[what] = z[k]
    def mk(d):
        dims = str(d["dims"])
        return DataArray(d["values"], dims=tuple(x for x in dims.split(",") if x), attrs={"units": str(d["units"])})
    state = {n: mk(d) for n, d in groups["state"].items()}
    state["time"] = datetime.datetime(2000, 1, 1)
    return state, {n: mk(d) for n, d in groups["tend"].items()}, {n: mk(d) for n, d in groups["diag"].items()}


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


# ---- the longwave drop-in class against the reference Fortran fed by an independent restatement of the reference's host layer
# (tests/golden/make_golden.py::reference_lw_class_cases; on this build's table blob -- synthetic while the data file is missing)
LW_CACHE_CLASSES = (("TestRRTMGLongwave", "column", {}),
                    ("TestRRTMGLongwaveWithClouds", "column", dict(cloud_optical_properties="single_cloud_type")),
                    ("TestRRTMGLongwaveWithExternalInterfaceTemperature", "column", dict(calculate_interface_temperature=False)),
                    ("TestRRTMGLongwaveMCICA", "3d", dict(mcica=True)))
LWCLASS_CASES = tuple(sorted(f[len("ref_lwclass_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("ref_lwclass_") and f.endswith(".npz")))


def load_lwclass_case(name):
    """-> (state of DataArrays, constructor kwargs, expected tendencies, expected diagnostics, fixture made on synthetic tables?).
    The four cases named after the reference's cache classes take their state from the cache fixture."""
    import json
    from climt_amd._sympl_compat import DataArray
    z = np.load(os.path.join(GOLDEN, "ref_lwclass_%s.npz" % name))
    groups = {"state": {}, "tend": {}, "diag": {}}
    for k in z.files:
        if "/" in k:
            grp, q, what = k.split("/")
            groups[grp].setdefault(q, {})[what] = z[k]
    def mk(d):
        return DataArray(d["values"], dims=tuple(x for x in str(d["dims"]).split(",") if x), attrs={"units": str(d["units"])})
    if groups["state"]:
        state = {n: mk(d) for n, d in groups["state"].items()}
        state["time"] = datetime.datetime(2000, 1, 1)
    else:
        state = load_cache_case(*name.rsplit("-", 1))[0]
    return (state, json.loads(str(z["kwargs"])), {n: mk(d) for n, d in groups["tend"].items()}, {n: mk(d) for n, d in groups["diag"].items()},
            bool(int(z["synthetic_tables"])))


def check_lwclass_case(name, make_component, tol=1e-9):
    """Run climt_amd.RRTMGLongwave(**kwargs)(state) (make_component(**kwargs) builds it) and compare every returned quantity."""
    state, kw, tend, diag, synthetic = load_lwclass_case(name)
    comp = make_component(**kw)
    assert comp._ctx.lw_tables_synthetic() == synthetic, "ref_lwclass_* fixtures were made on another table blob: python tests/golden/make_golden.py lwclass"
    np.random.seed(0)                                     # tests/test_components.py:148
    t, d = comp(state)
    assert set(t) == set(tend) and set(d) == set(diag)
    worst = 0.0
    for got, exp in ((t, tend), (d, diag)):
        for k in exp:
            assert got[k].attrs["units"].replace("degK", "K") == exp[k].attrs["units"].replace("degK", "K")
            assert set(got[k].dims) == set(exp[k].dims), (k, got[k].dims, exp[k].dims)
            g = np.transpose(got[k].values, [got[k].dims.index(x) for x in exp[k].dims])
            assert g.shape == exp[k].values.shape
            dd = maxdiff(g, exp[k].values)
            assert dd <= tol, (name, k, dd)
            worst = max(worst, dd)
    assert d["air_temperature_tendency_from_longwave"].values is t["air_temperature"].values or \
        np.array_equal(d["air_temperature_tendency_from_longwave"].values, t["air_temperature"].values)
    return worst


LWMR_CASES = ("maxrand", "maxrand_idrv", "maximum")


def load_lwmr_case(name):
    """Non-McICA maximum/random overlap (rtrnmr) fixture -> (inputs at the C-ABI boundary, expected LW outputs)."""
    from climt_amd.synthetic import make_columns
    z = np.load(os.path.join(GOLDEN, "ref_lwmr_%s.npz" % name))
    c = make_columns(40, 60, cloudy=True, seed=int(z["flag/seed"]))
    c["cldfr"] = np.ascontiguousarray(z["in/cldfr"])
    c.update(icld=int(z["flag/icld"]), iaer=0, inflg=2, iceflg=1, liqflg=1, idrv=int(z["flag/idrv"]))
    _check_pinned_inputs("ref_lwmr_" + name, c)
    return c, {k[3:]: z[k] for k in z.files if k.startswith("lw/")}


# ---- option-coverage fixtures (self-contained: the inputs are stored in the file) -----------------------------
OPT_CASES = tuple(sorted(f[len("ref_opt_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("ref_opt_") and f.endswith(".npz")))


def load_opt_case(name):
    """-> (spectrum 'sw' | 'lw', mcica flag, inputs at the C-ABI boundary, expected outputs of the reference Fortran)."""
    z = np.load(os.path.join(GOLDEN, "ref_opt_%s.npz" % name))
    c = {k[3:]: np.ascontiguousarray(z[k]) for k in z.files if k.startswith("in/")}
    flags = {k[5:]: z[k].item() for k in z.files if k.startswith("flag/")}
    mcica = bool(flags.pop("_mcica"))
    c.update(flags)
    spectrum = name.split("_")[0]
    exp = {k[3:]: z[k] for k in z.files if k.startswith(spectrum + "/")}
    return spectrum, mcica, c, exp


# ---- the live oracle on many columns: reference library (oracle/_ref) if it travelled, else the C restatement ----------
def _live_oracle_worker(args):
    """One host process (the reference Fortran keeps process-global state): SW and LW of one column chunk."""
    chunk, mcica, spectra = args
    sys.path.insert(0, ROOT)
    from oracle import ref_driver
    sw = lw = None
    if ref_driver.available("sw") and ref_driver.available("lw"):
        from tools.pack_tables import read_blob
        from tools.synth_lw_tables import fill_reference_from_blob
        if "sw" in spectra:
            sw = ref_driver.RefSW().fluxes(chunk, mcica=mcica)
        if "lw" in spectra:
            blob = read_blob(LW_DATA)
            rlw = ref_driver.RefLW(); rlw.init(fill_tables=lambda r: fill_reference_from_blob(r, blob))
            lw = rlw.fluxes(chunk, mcica=mcica)
        kind = "reference"
    else:
        from oracle.port_driver import PortLW, PortSW
        if "sw" in spectra:
            sw = PortSW().fluxes(chunk, mcica=mcica)
        if "lw" in spectra:
            lw = PortLW().fluxes(chunk, mcica=mcica)
        kind = "port"
    return ({k: sw[k] for k, _ in SW_OUT} if sw else None, {k: lw[k] for k, _ in LW_OUT} if lw else None, kind)


def live_oracle(c, mcica, chunk=128, procs=None, spectra=("sw", "lw"), timeout=900):
    """SW and LW outputs of the oracle for the columns of `c` (kissvec or clear sky: columns are independent), computed in
    column chunks on a pool of host processes -> (sw dict, lw dict, 'reference' | 'port'); a spectrum not asked for is None.
    (The reference Fortran `stop`s on inputs it refuses -- fractional clouds in the non-McICA shortwave, say -- which ends
    the worker process and would leave Pool.map waiting for ever: the wait is bounded.)"""
    import multiprocessing as mp
    from climt_amd.distributed import slice_columns
    ncol = c["play"].shape[1]
    c = {k: v for k, v in c.items() if k != "lat"}
    jobs = [(slice_columns(c, s, min(ncol, s + chunk)), mcica, tuple(spectra)) for s in range(0, ncol, chunk)]
    procs = procs or max(1, min(len(jobs), os.cpu_count() or 1, 32))
    with mp.get_context("spawn").Pool(procs) as pool:
        parts = pool.map_async(_live_oracle_worker, jobs).get(timeout=timeout)
    sw = {k: np.concatenate([p[0][k] for p in parts], axis=1) for k, _ in SW_OUT} if "sw" in spectra else None
    lw = {k: np.concatenate([p[1][k] for p in parts], axis=1) for k, _ in LW_OUT} if "lw" in spectra else None
    require_reference_oracle(parts[0][2])
    assert all(p[2] == parts[0][2] for p in parts)
    return sw, lw, parts[0][2]


def require_reference_oracle(kind):
    """The live oracle of the GPU tests is the reference Fortran itself (oracle/_ref, built in the container and shipped to
    the GPU box with the snapshot).  If the libraries did not travel the tests would silently fall back to the C restatement:
    that must be somebody's decision (RRTMG_TEST_ALLOW_PORT_ORACLE=1), not an accident."""
    if os.environ.get("RRTMG_TEST_ALLOW_PORT_ORACLE", "") in ("", "0"):
        assert kind == "reference", ("oracle/_ref/librrtmg_{sw,lw}_ref.so are not here: the live oracle would be the C restatement "
                                     "(oracle/), not the reference Fortran.  Build them (oracle/build_ref.sh) or set "
                                     "RRTMG_TEST_ALLOW_PORT_ORACLE=1 to compare with the restatement knowingly.")


# ---- every `stop` message of the reference has its own status code (include/rrtmg_hip.h) -----------------------------------
# (code, text of the reference's stop, spectrum, mcica, changes to the overcast_L60 case, poisoned shortwave table | None |
#  "host" = checked by the library's host code before any kernel runs: not part of the device-function emulation)
def _low_first_layer(c):
    p = np.array(c["play"]); p[0] = p[1] - 1.0   # pressure rising with height in the two lowest layers
    return dict(play=p, irng=0, permuteseed=3)


STOP_CASES = [
    (10, "PARTIAL CLOUD NOT ALLOWED", "sw", False, lambda c: dict(cldfr=np.where(c["cldfr"] > 0, 0.5, 0.0)), None),
    (11, "ICE RADIUS OUT OF BOUNDS", "sw", False, lambda c: dict(reice=np.full_like(c["reice"], 500.0)), None),
    (11, "ICE RADIUS OUT OF BOUNDS", "lw", True, lambda c: dict(reice=np.full_like(c["reice"], 4.0), iceflg=2, irng=0, permuteseed=1), None),
    (12, "LIQUID EFFECTIVE RADIUS OUT OF BOUNDS", "lw", False, lambda c: dict(reliq=np.full_like(c["reliq"], 1.0)), None),
    (12, "LIQUID EFFECTIVE RADIUS OUT OF BOUNDS", "sw", True, lambda c: dict(reliq=np.full_like(c["reliq"], 61.0), irng=0, permuteseed=1), None),
    (14, "KISSVEC SEED GENERATOR REQUIRES PMID FROM BOTTOM FOUR LAYERS", "sw", True, _low_first_layer, None),
    (14, "KISSVEC SEED GENERATOR REQUIRES PMID FROM BOTTOM FOUR LAYERS", "lw", True, _low_first_layer, None),
    (16, "INFLAG = 1 OPTION NOT AVAILABLE WITH MCICA", "sw", True, lambda c: dict(inflg=1, irng=0, permuteseed=1), "host"),
    (16, "INFLAG = 1 OPTION NOT AVAILABLE WITH MCICA", "lw", True, lambda c: dict(inflg=1, irng=0, permuteseed=1), None),
    (17, "ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS", "sw", False, lambda c: dict(reice=np.full_like(c["reice"], 200.0), iceflg=3), None),
    (17, "ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS", "lw", False, lambda c: dict(reice=np.full_like(c["reice"], 200.0), iceflg=3), None),
    (18, "ICE RADIUS TOO SMALL", "lw", False, lambda c: dict(reice=np.full_like(c["reice"], 5.0), iceflg=0), None),
    (20, "", "sw", False, lambda c: dict(inflg=1), "host"),      # no shortwave implementation of inflag 1 in RRTMG_SW (host check)
    (20, "", "sw", False, lambda c: dict(iceflg=0), None),       # ... nor of iceflag 0 (device check)
    (30, "ICE EXTINCTION LESS THAN 0.0", "sw", False, lambda c: dict(iceflg=2), ("sw/cld/extice2", -1.0)),
    (31, "ICE SSA GRTR THAN 1.0", "sw", False, lambda c: dict(iceflg=2), ("sw/cld/ssaice2", 1.5)),
    (32, "ICE SSA LESS THAN 0.0", "sw", True, lambda c: dict(iceflg=2, irng=0, permuteseed=1), ("sw/cld/ssaice2", -0.5)),
    (33, "ICE ASYM GRTR THAN 1.0", "sw", False, lambda c: dict(iceflg=2), ("sw/cld/asyice2", 1.5)),
    (34, "ICE ASYM LESS THAN 0.0", "sw", False, lambda c: dict(iceflg=3), ("sw/cld/asyice3", -0.5)),
    (35, "FDELTA LESS THAN 0.0", "sw", False, lambda c: dict(iceflg=3), ("sw/cld/fdlice3", -0.5)),
    (36, "FDELTA GT THAN 1.0", "sw", True, lambda c: dict(iceflg=3, irng=0, permuteseed=1), ("sw/cld/fdlice3", 1.5)),
    (37, "LIQUID EXTINCTION LESS THAN 0.0", "sw", False, lambda c: {}, ("sw/cld/extliq1", -1.0)),
    (38, "LIQUID SSA GRTR THAN 1.0", "sw", False, lambda c: {}, ("sw/cld/ssaliq1", 1.5)),
    (39, "LIQUID SSA LESS THAN 0.0", "sw", False, lambda c: {}, ("sw/cld/ssaliq1", -0.5)),
    (40, "LIQUID ASYM GRTR THAN 1.0", "sw", True, lambda c: dict(irng=0, permuteseed=1), ("sw/cld/asyliq1", 1.5)),
    (41, "LIQUID ASYM LESS THAN 0.0", "sw", False, lambda c: {}, ("sw/cld/asyliq1", -0.5)),
]


def poisoned_sw_blob(path, name, value):
    """The shortwave data blob with every entry of ONE table replaced by `value` (the cloud-optics checks of cldprop_sw /
    cldprmc_sw look at numbers interpolated from these tables: with the shipped data they can never fail)."""
    from tools.pack_tables import Blob, read_blob
    out = Blob()
    hit = False
    for n, arr in read_blob(SW_DATA).items():
        a = np.array(arr)
        if n == name:
            a, hit = np.full_like(a, value), True
        out.add(n, a)
    assert hit, name
    out.write(path)
    return path


def run_stop_case(ctx, case, tmp_path, base=None):
    """Runs one STOP_CASES entry on `ctx` (Context or EmuContext; its shortwave tables are re-initialised from the poisoned
    blob when the case has one -- and from the shipped blob again afterwards) and returns the RRTMGError it raised."""
    import pytest
    from climt_amd._lib import RRTMGError
    code, text, which, mcica, change, poison = case
    c = dict(base) if base is not None else dict(load_ref_case("overcast_L60")[0])
    c.update(change(c))
    poison = None if poison == "host" else poison
    if poison:
        ctx.sw_init(CPDAIR, blob=poisoned_sw_blob(os.path.join(str(tmp_path), "poisoned_sw.bin"), *poison))
    try:
        with pytest.raises(RRTMGError) as e:
            (ctx.sw_fluxes if which == "sw" else ctx.lw_fluxes)(c, mcica=mcica)
    finally:
        if poison:
            ctx.sw_init(CPDAIR)
    assert e.value.code == code, (e.value.code, str(e.value))
    return e.value
