"""ctypes binding of librrtmg_hip.so (include/rrtmg_hip.h) -- replaces climt's Cython shims
_rrtmg_sw.pyx / _rrtmg_lw.pyx (climt/_components/rrtmg/{sw,lw}/).

There is no CPU fallback: importing works anywhere (so property dictionaries can be inspected),
but creating a Context without the built library or without a GPU raises.
"""
import ctypes as C
import functools
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RRTMG_HIP_LIB") or os.path.join(HERE, "_lib", "librrtmg_hip.so")
SW_DATA = os.path.join(HERE, "data", "rrtmg_sw_data.bin")
LW_DATA = os.path.join(HERE, "data", "rrtmg_lw_data.bin")

_vp, _i32, _f64 = C.c_void_p, C.c_int32, C.c_double


class RRTMGError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rrtmg_hip error %d: %s" % (code, msg))
        self.code = code


# unit factors the library applies to host arrays on the device (include/rrtmg_hip.h: rrtmg_sw_args, last four fields)
_SCALES = ("pressure_scale", "water_path_scale", "h2o_mul", "h2o_div")


class SwArgs(C.Structure):
    _fields_ = ([(n, _i32) for n in ("ncol nlay memspace mcica icld iaer inflgsw iceflgsw liqflgsw dyofyr isolvar "
                                     "irng permuteseed shard_col0 shard_ncol struct_size").split()]
                + [(n, _f64) for n in "adjes scon solcycfrac".split()]
                + [(n, _vp) for n in ("bndsolvar indsolvar play plev tlay tlev tsfc h2ovmr o3vmr co2vmr ch4vmr n2ovmr o2vmr "
                                      "asdir asdif aldir aldif coszen cldfr taucld ssacld asmcld fsfcld cicewp cliqwp reice "
                                      "reliq tauaer ssaaer asmaer ecaer cldfmcl swuflx swdflx swhr swuflxc swdflxc swhrc").split()]
                + [(n, _f64) for n in _SCALES])


class LwArgs(C.Structure):
    _fields_ = ([(n, _i32) for n in ("ncol nlay memspace mcica icld idrv inflglw iceflglw liqflglw irng permuteseed "
                                     "shard_col0 shard_ncol struct_size").split()]
                + [(n, _vp) for n in ("play plev tlay tlev tsfc h2ovmr o3vmr co2vmr ch4vmr n2ovmr o2vmr cfc11vmr cfc12vmr "
                                      "cfc22vmr ccl4vmr emis cldfr taucld cicewp cliqwp reice reliq tauaer cldfmcl "
                                      "uflx dflx hr uflxc dflxc hrc duflx_dt duflxc_dt").split()]
                + [(n, _f64) for n in _SCALES])


SLAB_IN = ("sw_down lw_down sw_up lw_up lh sh up_heat_soil heat_flux_sea_ice sea_water_dens surf_dens heat_cap_soil surf_therm_cap "
           "ocean_mix_thick soil_layer_thick ocean_heat_transport").split()


class SlabArgs(C.Structure):
    """mirrors `rrtmg_slab_args` (include/rrtmg_hip.h), field for field"""
    _fields_ = [(n, _vp) for n in ("sw_down lw_down sw_up lw_up lh sh area_type up_heat_soil heat_flux_sea_ice sea_water_dens surf_dens "
                                   "heat_cap_soil surf_therm_cap ocean_mix_thick soil_layer_thick ocean_heat_transport tend_ts depth").split()]


_lib = None


def load_library():
    """Load librrtmg_hip.so; raises ImportError (as climt does for its missing Fortran extension,
    lw/component.py:253-257) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("librrtmg_hip.so has not been built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.rrtmg_hip_last_error.restype = C.c_char_p
    lib.rrtmg_hip_last_error.argtypes = [_vp]
    lib.rrtmg_hip_version.restype = C.c_char_p
    lib.rrtmg_hip_stream.restype = _vp
    lib.rrtmg_hip_stream.argtypes = [_vp]
    lib.rrtmg_hip_create.argtypes = [C.POINTER(_vp), C.c_int]
    lib.rrtmg_hip_destroy.argtypes = [_vp]
    lib.rrtmg_hip_set_constants.argtypes = [_vp] + [_f64] * 10
    lib.rrtmg_hip_sw_init.argtypes = [_vp, _f64, C.c_char_p]
    lib.rrtmg_hip_lw_init.argtypes = [_vp, _f64, C.c_char_p]
    lib.rrtmg_hip_sw_fluxes.argtypes = [_vp, C.POINTER(SwArgs)]
    lib.rrtmg_hip_lw_fluxes.argtypes = [_vp, C.POINTER(LwArgs)]
    lib.rrtmg_hip_get_table.restype = C.c_long
    lib.rrtmg_hip_get_table.argtypes = [_vp, C.c_char_p, _vp, C.c_long]
    lib.rrtmg_hip_lw_tables_synthetic.argtypes = [_vp]
    lib.rrtmg_hip_synchronize.argtypes = [_vp]
    lib.rrtmg_hip_set_deferred.argtypes = [_vp, C.c_int]
    lib.rrtmg_hip_stream_wait.argtypes = [_vp, _vp]
    lib.rrtmg_hip_interface_values.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
    lib.rrtmg_hip_elementwise.argtypes = [_vp, C.c_int, C.c_long, _vp, _vp, _f64, _f64, _vp]
    lib.rrtmg_hip_ab_step.argtypes = [_vp, C.c_long, C.c_int, _vp, C.POINTER(_vp), C.POINTER(_f64), _f64, _vp]
    lib.rrtmg_hip_order_streams.argtypes = [_vp, C.c_int]
    lib.rrtmg_hip_zenith_angle.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _f64, _vp]
    lib.rrtmg_hip_slab_surface.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(SlabArgs)]
    lib.rrtmg_hip_solar_insolation.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _f64, _f64, _f64, _f64, _vp, _vp]
    lib.rrtmg_hip_kernel_ms.argtypes = [_vp, C.c_int, C.POINTER(C.c_double)]
    lib.rrtmg_hip_kernel_launches.argtypes = [_vp, C.c_int]
    lib.rrtmg_hip_set_column_sort.argtypes = [_vp, C.c_int]
    lib.rrtmg_hip_copy_blocks.argtypes = [_vp, C.c_int, _vp, C.c_long, C.c_long, _vp, _vp, _vp]
    lib.rrtmg_hip_mcica_mask.argtypes = [_vp] + [C.c_int] * 6 + [_vp] * 3
    _lib = lib
    return lib


# the ten constants in the order of rrtmg_sw_set_constants (rrtmg_sw_c_binder.f90:19-46)
CONSTANT_NAMES = ("pi", "grav", "planck", "boltz", "clight", "avogad", "alosmt", "gascon", "sbcnst", "secdy")

# boundary-level names (left) -> SwArgs / LwArgs field (right)
_SW_FIELDS = dict(play="play", plev="plev", tlay="tlay", tlev="tlev", tsfc="tsfc", h2o="h2ovmr", o3="o3vmr", co2="co2vmr",
                  ch4="ch4vmr", n2o="n2ovmr", o2="o2vmr", asdir="asdir", asdif="asdif", aldir="aldir", aldif="aldif",
                  coszen="coszen", cldfr="cldfr", taucld="taucld", ssacld="ssacld", asmcld="asmcld", fsfcld="fsfcld",
                  cicewp="cicewp", cliqwp="cliqwp", reice="reice", reliq="reliq", tauaer="tauaer", ssaaer="ssaaer",
                  asmaer="asmaer", ecaer="ecaer", cldfmcl="cldfmcl", bndsolvar="bndsolvar", indsolvar="indsolvar")
_LW_FIELDS = dict(play="play", plev="plev", tlay="tlay", tlev="tlev", tsfc="tsfc", h2o="h2ovmr", o3="o3vmr", co2="co2vmr",
                  ch4="ch4vmr", n2o="n2ovmr", o2="o2vmr", cfc11="cfc11vmr", cfc12="cfc12vmr", cfc22="cfc22vmr", ccl4="ccl4vmr",
                  emis="emis", cldfr="cldfr", taucld="taucld", cicewp="cicewp", cliqwp="cliqwp", reice="reice", reliq="reliq",
                  tauaer="tauaer", cldfmcl="cldfmcl")
_SW_FLAGS = dict(icld="icld", iaer="iaer", inflg="inflgsw", iceflg="iceflgsw", liqflg="liqflgsw", dyofyr="dyofyr",
                 isolvar="isolvar", irng="irng", permuteseed="permuteseed", shard_col0="shard_col0", shard_ncol="shard_ncol")
_LW_FLAGS = dict(icld="icld", idrv="idrv", inflg="inflglw", iceflg="iceflglw", liqflg="liqflglw", irng="irng",
                 permuteseed="permuteseed", shard_col0="shard_col0", shard_ncol="shard_ncol")
SW_OUT = (("swuflx", 1), ("swdflx", 1), ("swhr", 0), ("swuflxc", 1), ("swdflxc", 1), ("swhrc", 0))
LW_OUT = (("uflx", 1), ("dflx", 1), ("hr", 0), ("uflxc", 1), ("dflxc", 1), ("hrc", 0))


def source_hash():
    """The hash of the sources the loaded library was built from (rrtmg_hip_version(): "... src:<16 hex digits>")."""
    v = load_library().rrtmg_hip_version().decode()
    return v.split("src:")[1].strip() if "src:" in v else "unknown"


def _locked(fn):
    """Context methods that enter the library hold the context's lock: ctypes releases the GIL for the duration of a
    call, and a context -- its staging buffers, work-buffer map, error string, streams -- is shared by every component of
    the process on that device (climt_amd.rrtmg.common.make_context), so two Python threads must not be inside it at once."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        with self._lock:
            return fn(self, *args, **kwargs)
    return wrapper


class Context:
    """One librrtmg_hip context: constants, device tables, work buffers, HIP streams.  Calls are serialised per context
    (a re-entrant lock), so components sharing it may be driven from several Python threads."""

    def __init__(self, device=0):
        self.lib = load_library()
        self._lock = threading.RLock()
        self.deferred = False
        h = _vp()
        rc = self.lib.rrtmg_hip_create(C.byref(h), int(device))
        self.h = h
        self.device = device
        if rc:
            msg = self.lib.rrtmg_hip_last_error(self.h).decode() if self.h else "context creation failed"
            raise RRTMGError(rc, msg)

    def close(self):
        if getattr(self, "h", None):
            self.lib.rrtmg_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise RRTMGError(rc, self.lib.rrtmg_hip_last_error(self.h).decode())

    @_locked
    def set_constants(self, **k):
        self._ck(self.lib.rrtmg_hip_set_constants(self.h, *[float(k[n]) for n in CONSTANT_NAMES]))

    @_locked
    def sw_init(self, cpdair, blob=None):
        blob = blob or os.environ.get("RRTMG_HIP_SW_DATA") or SW_DATA
        key = ("sw", float(cpdair), blob)
        if getattr(self, "_sw_key", None) != key:       # (components sharing the context initialise the tables once)
            self._ck(self.lib.rrtmg_hip_sw_init(self.h, float(cpdair), blob.encode()))
            self._sw_key = key

    @_locked
    def lw_init(self, cpdair, blob=None):
        blob = blob or os.environ.get("RRTMG_HIP_LW_DATA") or LW_DATA      # (a packed table file elsewhere: tools/ingest_lw_data.sh)
        key = ("lw", float(cpdair), blob)
        if getattr(self, "_lw_key", None) != key:
            self._ck(self.lib.rrtmg_hip_lw_init(self.h, float(cpdair), blob.encode()))
            self._lw_key = key

    def lw_tables_synthetic(self):
        return bool(self.lib.rrtmg_hip_lw_tables_synthetic(self.h))

    @property
    def stream(self):
        return self.lib.rrtmg_hip_stream(self.h)

    @_locked
    def kernel_ms(self, which, cloudy=False):
        """HIP-event duration (ms) of a solve kernel in the last call, summed over the call's column chunks (one launch
        each): which = 'sw' | 'lw'; cloudy selects the kernel that handles the cloudy tiles (sw_solve_cloudy_kernel /
        lw_solve_all_kernel<true,..>) instead of the clear-sky one."""
        ms = C.c_double(0.0)
        self._ck(self.lib.rrtmg_hip_kernel_ms(self.h, (0 if which == "sw" else 1) + (2 if cloudy else 0), C.byref(ms)))
        return ms.value

    @_locked
    def kernel_launches(self, which, cloudy=False):
        """Launches (column chunks) of that solve kernel in the last call; kernel_ms is their sum."""
        return int(self.lib.rrtmg_hip_kernel_launches(self.h, (0 if which == "sw" else 1) + (2 if cloudy else 0)))

    @_locked
    def copy_blocks(self, desc_ptr, nblk, max_rows, max_cols, src, dst, stream=None):
        """rrtmg_hip_copy_blocks: nblk strided 2-d block copies (device pointers; desc = device int64[nblk][6]) on `stream`."""
        self._ck(self.lib.rrtmg_hip_copy_blocks(self.h, int(nblk), _vp(desc_ptr), int(max_rows), int(max_cols), _vp(src), _vp(dst), _vp(stream)))

    @_locked
    def synchronize(self):
        """Wait for all enqueued work; in deferred mode this is where device-side errors are raised."""
        self._ck(self.lib.rrtmg_hip_synchronize(self.h))

    @_locked
    def stream_wait(self, other_stream):
        """`other_stream` (hipStream_t) waits, on the device, for everything enqueued so far on this context's streams."""
        self._ck(self.lib.rrtmg_hip_stream_wait(self.h, _vp(other_stream)))

    # -- glue of the device-resident step (device pointers) ------------------------------------------
    @_locked
    def interface_values(self, ncol, nlay, mid, surf, pmid, pint, out):
        self._ck(self.lib.rrtmg_hip_interface_values(self.h, int(ncol), int(nlay), mid, surf, pmid, pint, out))

    @_locked
    def elementwise(self, op, n, a, out, b=None, alpha=1.0, beta=1.0):
        """op: 'axpby' out = alpha*a (+ beta*b), 'cos' out = cos(a), 'muldiv' out = a*alpha/beta"""
        self._ck(self.lib.rrtmg_hip_elementwise(self.h, {"axpby": 0, "cos": 1, "muldiv": 2}[op], int(n), a, b, float(alpha), float(beta), out))

    @_locked
    def ab_step(self, n, x, tendencies, weights, dt, out):
        k = len(tendencies)
        f = (_vp * k)(*tendencies)
        w = (_f64 * k)(*weights)
        self._ck(self.lib.rrtmg_hip_ab_step(self.h, int(n), k, x, f, w, float(dt), out))

    @_locked
    def order_streams(self, direction):
        self._ck(self.lib.rrtmg_hip_order_streams(self.h, int(direction)))

    @_locked
    def slab_surface_device(self, ncol, ptrs, area_type, tend_ts, depth):
        """rrtmg_hip_slab_surface on device pointers: `ptrs` maps SLAB_IN names to device addresses."""
        a = SlabArgs()
        for k in SLAB_IN:
            setattr(a, k, int(ptrs[k]))
        a.area_type, a.tend_ts, a.depth = int(area_type), int(tend_ts), int(depth)
        self._ck(self.lib.rrtmg_hip_slab_surface(self.h, int(ncol), 1, C.byref(a)))

    @_locked
    def zenith_angle(self, lat_deg, lon_deg, julian_centuries, out=None, memspace=0, ncol=None):
        """Zenith angle (radians) of every column; host arrays, or device pointers with memspace=1 (then `ncol`)."""
        if memspace:
            self._ck(self.lib.rrtmg_hip_zenith_angle(self.h, int(ncol), 1, int(lat_deg), int(lon_deg), float(julian_centuries), int(out)))
            return out
        lat = np.ascontiguousarray(lat_deg, dtype=np.float64)
        lon = np.ascontiguousarray(lon_deg, dtype=np.float64)
        z = np.empty(lat.shape) if out is None else out
        self._ck(self.lib.rrtmg_hip_zenith_angle(self.h, lat.size, 0, lat.ctypes.data, lon.ctypes.data, float(julian_centuries), z.ctypes.data))
        return z

    @_locked
    def solar_insolation(self, lat, lon, sin_delta, cos_delta, fractional_day, irradiance):
        """(zenith angle, insolation) of every column (host arrays): per-column part of BergerSolarInsolation."""
        lat = np.ascontiguousarray(lat, dtype=np.float64)
        lon = np.ascontiguousarray(lon, dtype=np.float64)
        z, s = np.empty(lat.shape), np.empty(lat.shape)
        self._ck(self.lib.rrtmg_hip_solar_insolation(self.h, lat.size, 0, lat.ctypes.data, lon.ctypes.data, float(sin_delta), float(cos_delta),
                                                     float(fractional_day), float(irradiance), z.ctypes.data, s.ctypes.data))
        return z, s

    @_locked
    def slab_surface(self, area_type, **arrays):
        """Kernel of climt SlabSurface on host arrays: -> (surface temperature tendency, slab depth).  `arrays`: SLAB_IN."""
        a = SlabArgs()
        keep = [np.ascontiguousarray(area_type, dtype=np.int32)]
        a.area_type = keep[0].ctypes.data
        n = keep[0].size
        for k in SLAB_IN:
            v = np.ascontiguousarray(arrays[k], dtype=np.float64)
            assert v.size == n, k
            keep.append(v)
            setattr(a, k, v.ctypes.data)
        tend, depth = np.empty(n), np.empty(n)
        a.tend_ts, a.depth = tend.ctypes.data, depth.ctypes.data
        self._ck(self.lib.rrtmg_hip_slab_surface(self.h, n, 0, C.byref(a)))
        return tend, depth

    @_locked
    def set_deferred(self, on=True):
        """Device-resident (memspace=1) calls return after enqueueing; SW and LW overlap on two streams.  Returns the
        previous setting, so that whoever switches it on for the lifetime of an object can restore it (the context is
        shared: a memspace=1 caller that expects per-call synchronisation and error checks must get them back)."""
        prev = self.deferred
        self._ck(self.lib.rrtmg_hip_set_deferred(self.h, 1 if on else 0))
        self.deferred = bool(on)
        return prev

    @_locked
    def set_column_sort(self, on=True):
        """Opt-in internal column order of device-resident calls with clouds: cloud-free columns first (rrtmg_hip_set_column_sort)."""
        self._ck(self.lib.rrtmg_hip_set_column_sort(self.h, 1 if on else 0))

    @_locked
    def get_table(self, name):
        n = self.lib.rrtmg_hip_get_table(self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n)
        self.lib.rrtmg_hip_get_table(self.h, name.encode(), out.ctypes.data, n)
        return out

    # -- host-pointer calls: `inp` maps boundary names to numpy arrays (see _SW_FIELDS) -------
    def _fill(self, a, inp, fields, flags, keep):
        for k, f in flags.items():
            if k in inp:
                setattr(a, f, int(inp[k]))
        for k in _SCALES:
            if inp.get(k):
                setattr(a, k, float(inp[k]))
        for k, f in fields.items():
            v = inp.get(k)
            if v is None:
                continue
            if isinstance(v, (int, np.integer)):      # raw device pointer
                setattr(a, f, int(v))
            else:
                arr = np.ascontiguousarray(v, dtype=np.float64)
                keep.append(arr)
                setattr(a, f, arr.ctypes.data)

    @_locked
    def sw_fluxes(self, inp, mcica=False, out=None, memspace=0):
        nlay, ncol = (inp["nlay"], inp["ncol"]) if memspace else inp["play"].shape
        a = SwArgs()
        a.struct_size = C.sizeof(SwArgs)
        keep = []
        a.ncol, a.nlay, a.memspace, a.mcica = int(ncol), int(nlay), int(memspace), int(bool(mcica))
        a.icld, a.inflgsw, a.iceflgsw, a.liqflgsw, a.dyofyr = 1, 2, 1, 1, 1
        a.adjes, a.scon, a.solcycfrac = float(inp.get("adjes", 1.0)), float(inp.get("scon", 1367.0)), float(inp.get("solcycfrac", 0.0))
        self._fill(a, inp, _SW_FIELDS, _SW_FLAGS, keep)
        if out is None:
            out = {k: np.zeros((nlay + lev, ncol)) for k, lev in SW_OUT}
        for k, _ in SW_OUT:
            v = out[k]
            setattr(a, k, int(v) if isinstance(v, (int, np.integer)) else v.ctypes.data)
        self._ck(self.lib.rrtmg_hip_sw_fluxes(self.h, C.byref(a)))
        return out

    @_locked
    def lw_fluxes(self, inp, mcica=False, out=None, memspace=0):
        nlay, ncol = (inp["nlay"], inp["ncol"]) if memspace else inp["play"].shape
        a = LwArgs()
        a.struct_size = C.sizeof(LwArgs)
        keep = []
        a.ncol, a.nlay, a.memspace, a.mcica = int(ncol), int(nlay), int(memspace), int(bool(mcica))
        a.icld, a.inflglw, a.iceflglw, a.liqflglw = 1, 2, 1, 1
        self._fill(a, inp, _LW_FIELDS, _LW_FLAGS, keep)
        if out is None:
            out = {k: np.zeros((nlay + lev, ncol)) for k, lev in LW_OUT}
            if a.idrv:
                out["duflx_dt"] = np.zeros((nlay + 1, ncol))
                out["duflxc_dt"] = np.zeros((nlay + 1, ncol))
        for k in out:
            v = out[k]
            setattr(a, k, int(v) if isinstance(v, (int, np.integer)) else v.ctypes.data)
        self._ck(self.lib.rrtmg_hip_lw_fluxes(self.h, C.byref(a)))
        return out

    @_locked
    def mcica_mask(self, which, play, cldfrac, icld, permuteseed, irng):
        nlay, ncol = play.shape
        nsub = 112 if which == "sw" else 140
        out = np.zeros((nlay, ncol, nsub))
        p = np.ascontiguousarray(play, dtype=np.float64)
        c = np.ascontiguousarray(cldfrac, dtype=np.float64)
        self._ck(self.lib.rrtmg_hip_mcica_mask(self.h, 0 if which == "sw" else 1, ncol, nlay, int(icld), int(permuteseed),
                                               int(irng), p.ctypes.data, c.ctypes.data, out.ctypes.data))
        return out
