"""Keyword option -> RRTMG integer flag maps (same keys and values as
climt/_components/rrtmg/rrtmg_common.py:7-59) and the shared library context."""
import weakref

import numpy as np

from .._lib import CONSTANT_NAMES, Context
from .._sympl_compat import get_constant

rrtmg_cloud_overlap_method_dict = {"clear_only": 0, "random": 1, "maximum_random": 2, "maximum": 3}
rrtmg_cloud_props_dict = {"direct_input": 0, "single_cloud_type": 1, "liquid_and_ice_clouds": 2}
rrtmg_cloud_ice_props_dict = {"ebert_curry_one": 0, "ebert_curry_two": 1, "key_streamer_manual": 2, "fu": 3}
rrtmg_cloud_liquid_props_dict = {"radius_independent_absorption": 0, "radius_dependent_absorption": 1}
rrtmg_aerosol_input_dict = {"no_aerosol": 0, "ecmwf": 6, "all_aerosol_properties": 10}
rrtmg_random_number_dict = {"kissvec": 0, "mersenne_twister": 1}


def physical_constants():
    """The ten constants climt hands to rrtmg[_sw]_set_constants (lw/component.py:298-309)."""
    vals = (
        np.pi,
        get_constant("gravitational_acceleration", "m/s^2"),
        get_constant("planck_constant", "erg s"),
        get_constant("boltzmann_constant", "erg K^-1"),
        get_constant("speed_of_light", "cm s^-1"),
        get_constant("avogadro_constant", "mole^-1"),
        get_constant("loschmidt_constant", "cm^-3"),
        get_constant("universal_gas_constant", "erg mol^-1 K^-1"),
        get_constant("stefan_boltzmann_constant", "W cm^-2 K^-4"),
        get_constant("seconds_per_day", "dimensionless"),
    )
    return dict(zip(CONSTANT_NAMES, vals))


_shared = {}


def make_context(device):
    """The librrtmg_hip context of `device` with the constants set -- ONE per device, shared by every component of the
    process (tables, streams and work buffers once, not once per component instance; the reference likewise keeps one set of
    module-global tables).  Flags travel with each call, so sharing is invisible to the components."""
    ctx = _shared.get(device)
    if ctx is None or not getattr(ctx, "h", None):
        ctx = _shared[device] = Context(device)
        ctx.set_constants(**physical_constants())
    return ctx


class OutputPool:
    """Output buffers of earlier calls that NOBODY can see any more are handed out again.

    The reference allocates its outputs afresh on every call (initialize_numpy_arrays_with_properties), and so did this
    package -- but a fresh np.zeros array has no pages yet: the copy of the results into it takes a page fault per 4 KiB,
    3-4 ms per LW+SW call at 8192 columns x 60 levels, a quarter of the drop-in call.  A buffer whose every array view has
    been garbage-collected cannot be observed by anyone, so writing the next call's results into it is indistinguishable
    from a fresh array -- except that its pages are mapped.  A caller that keeps every result keeps getting fresh arrays.
    Only for outputs the library overwrites completely (the radiation components').

    Liveness is tracked explicitly, not through reference counts (whose values are an interpreter detail): the memory
    belongs to a `bytearray` (or a page-locked ctypes byte array); each hand-out wraps it in a NEW root array
    (`np.frombuffer`), and what the caller receives
    -- and every slice, reshape or DataArray made from it -- is a view whose `.base` chain ends at that root (numpy
    collapses view chains to the first array whose own base is not an ndarray).  `weakref.finalize` on the root returns
    the bytearray to the free list when the last such view has died; an interpreter that collects later only delays the
    re-use."""

    def __init__(self, keep=4):
        self._free, self._keep, self._out = {}, keep, 0

    def _release(self, key, backing):
        self._out -= 1
        lst = self._free.setdefault(key, [])
        if len(lst) < self._keep:
            lst.append(backing)

    def zeros_like_fresh(self, name, shape):
        shape = tuple(int(n) for n in shape)
        key = (name, shape)
        lst = self._free.get(key)
        if lst:
            backing = lst.pop()
        else:
            backing = None
            if self._out < 8 * self._keep:
                # page-locked when the HIP runtime is there: the library's device-to-host copies land in it directly.  (A caller
                # that keeps every result gets pageable arrays beyond the first few: page-locked memory is not for archives.)
                try:
                    from .._hip import pinned_buffer
                    backing = pinned_buffer(8 * int(np.prod(shape)))
                    np.frombuffer(backing, dtype=np.float64)[:] = 0.0
                except Exception:
                    backing = None
            if backing is None:
                backing = bytearray(8 * int(np.prod(shape)))      # zero-filled, pages touched
        root = np.frombuffer(backing, dtype=np.float64)
        weakref.finalize(root, self._release, key, backing)
        self._out += 1
        return root.reshape(shape)


def _scale_into(values, factor, divisor, out):
    np.multiply(values, factor, out=out)
    if divisor is not None:
        np.divide(out, divisor, out=out)


class InputStaging:
    """Reusable host buffers for the inputs a call has to FORM before it can hand them over -- unit conversions (Pa -> mbar,
    kg m^-2 -> g m^-2), the mass -> volume mixing ratio of water vapour: five 4 MB products per component call at 8192
    columns x 60 levels, each a pass over memory (0.4 ms on one core), a quarter of the drop-in call when formed one after
    the other into fresh arrays as the reference does.  Here they are formed concurrently (numpy releases the interpreter
    lock inside a ufunc) into buffers that are kept, page-locked when the HIP runtime is there (the upload then runs at the
    link rate), plain numpy otherwise.  Same operations per element, so the same values.  Nothing outlives the call that
    filled the buffers: array_call only reads its inputs."""
    _workers = None

    def __init__(self):
        self._bufs, self._pending = {}, []

    def array(self, name, shape):
        key = (name, tuple(int(n) for n in shape))
        buf = self._bufs.get(key)
        if buf is None:
            try:
                from .._hip import pinned_empty
                buf = pinned_empty(key[1])
            except Exception:
                buf = np.empty(key[1])
            self._bufs[key] = buf
        return buf

    def scaled(self, name, values, factor, divisor=None, pieces=1):
        """values * factor (/ divisor) into this staging's buffer `name`, formed in the background: wait() before reading."""
        if InputStaging._workers is None:
            from concurrent.futures import ThreadPoolExecutor
            InputStaging._workers = ThreadPoolExecutor(max_workers=6, thread_name_prefix="rrtmg-stage")
        buf = self.array(name, values.shape)
        if pieces > 1 and values.ndim >= 1 and values.shape[0] >= pieces:
            edges = np.linspace(0, values.shape[0], pieces + 1).astype(int)
            for a, b in zip(edges[:-1], edges[1:]):
                self._pending.append(InputStaging._workers.submit(_scale_into, values[a:b], factor, divisor, buf[a:b]))
        else:
            self._pending.append(InputStaging._workers.submit(_scale_into, values, factor, divisor, buf))
        return buf

    def wait(self):
        pending, self._pending = self._pending, []
        for f in pending:
            f.result()


# Inputs whose unit factor the library applies on the device after the upload (include/rrtmg_hip.h: pressure_scale,
# water_path_scale) instead of this package on the host: the stand-in for sympl's extraction (_sympl_compat._extract) hands
# them over unconverted under name + "@raw" with the factor in state["_unit_factors"]; state[name] is then absent (nothing on the
# host can read it in the wrong unit).  (With the real sympl the arrays arrive converted under their names: nothing to do.)
_PRESSURES = ("air_pressure", "air_pressure_on_interface_levels")
_WATER_PATHS = ("mass_content_of_cloud_ice_in_atmosphere_layer", "mass_content_of_cloud_liquid_water_in_atmosphere_layer")
UNIT_FACTOR_ON_DEVICE = _PRESSURES + _WATER_PATHS
RAW = "@raw"


def library_scales(state):
    """-> (scales, arrays): the scale arguments of a host-pointer library call for the raw state of array_call, and the four
    arrays they apply to.  pressure_scale / water_path_scale come from the unit factors left unapplied -- one factor per pair
    of arrays: a state with, say, the two pressures in different units gets the unconverted ones converted here, into
    state[name] -- and h2o_mul / h2o_div are the water-vapour mass -> volume mixing ratio (util.py:86: q * 28.964 / 18.02)."""
    factors = state.get("_unit_factors") or {}
    scales = {"h2o_mul": 28.964, "h2o_div": 18.02}
    arrays = {}
    for names, key in ((_PRESSURES, "pressure_scale"), (_WATER_PATHS, "water_path_scale")):
        raw = [state.get(n + RAW) for n in names]
        f = [factors.get(n) for n in names]
        if raw[0] is not None and raw[1] is not None and f[0] == f[1]:
            scales[key] = f[0]
            arrays.update(zip(names, raw))
        else:
            for n, r, fac in zip(names, raw, f):
                if r is not None:
                    state[n] = r * fac      # (a new array in the unit input_properties declares; the caller's is untouched)
                arrays[n] = state[n]
    return scales, arrays


def output_arrays(pool, output_properties, raw_input_state, input_properties):
    """initialize_numpy_arrays_with_properties with recycling (OutputPool): shapes from the dims of the extracted inputs."""
    lengths = {}
    for name, prop in input_properties.items():
        v = raw_input_state.get(name) if hasattr(raw_input_state, "get") else None
        if v is None and hasattr(raw_input_state, "get"):
            v = raw_input_state.get(name + RAW)      # (an input handed over unconverted: same shape)
        if isinstance(v, np.ndarray):
            for dim, n in zip(prop.get("dims", ()), v.shape):
                lengths[dim] = n
    out = {}
    for name, prop in output_properties.items():
        dims = prop.get("dims")
        if dims is None:
            dims = input_properties[name]["dims"]
        out[name] = pool.zeros_like_fresh(name, [lengths[d] for d in dims])
    return out
