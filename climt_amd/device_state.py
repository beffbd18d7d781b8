"""A model state that lives in HBM, and the radiation step on it.

The reference's model loop (examples/gmd_aquaplanet.py:61-104) hands every component a host state and gets host arrays
back; with the kernels on the GPU that is 0.2-0.5 GB over PCIe per radiation call.  Here the state is uploaded ONCE:

    dstate = DeviceState.from_host(state, [sun, sw, lw, slab])       # component units and [levels][columns] layout
    stepper = DeviceAdamsBashforth(sw, lw, slab)                     # the same component instances as on the host
    for ...:
        dstate.update(sun(dstate))                                   # components recognise a DeviceState
        diagnostics, dstate = stepper(dstate, timestep)              # SW || LW -> tendency sum -> slab -> Adams-Bashforth
        dstate.update(diagnostics)
        dstate["time"] += timestep
    olr = dstate.download("upwelling_longwave_flux_in_air")          # only what is looked at comes back

`component(dstate)` returns (tendencies, diagnostics) -- or diagnostics -- whose values are DeviceQuantity handles; the
host work of `array_call` between the kernels (interface temperatures, q -> volume mixing ratio, cos(zenith), tendency
sums, the Adams-Bashforth update) runs as small kernels on the same streams (rrtmg_hip_interface_values /
_elementwise / _ab_step), shortwave and longwave overlap on the context's two streams, and nothing is copied until
`download`.  One library context (tables of both spectra) is shared by all components of a DeviceState.
"""
import datetime

import numpy as np

from . import _hip
from . import _sympl_compat as _sc
from ._lib import SLAB_IN
from .rrtmg.common import make_context


class DeviceQuantity:
    """A quantity in HBM: float64 (or int32) array in a component's layout -- dims like ['mid_levels', '*'], the wildcard
    being the flattened horizontal grid -- with its units."""

    def __init__(self, buf, shape, dims, units):
        self.buf, self.shape, self.dims, self.units = buf, tuple(int(s) for s in shape), tuple(dims), units

    @property
    def ptr(self):
        return self.buf.ptr

    @property
    def size(self):
        return int(np.prod(self.shape))

    def __repr__(self):
        return "DeviceQuantity(%s, dims=%s, units=%s)" % (self.shape, self.dims, self.units)


def _units_of(da):
    return getattr(da, "attrs", {}).get("units", "")


def _convert(values, src, dst):
    if hasattr(_sc, "convert_units"):
        return _sc.convert_units(values, src, dst)
    return _sc.DataArray(values, attrs={"units": src}).to_units(dst).values      # real sympl / xarray


def _same_units(a, b):
    if hasattr(_sc, "_canon"):
        return _sc._canon(a) == _sc._canon(b)
    return a == b


class DeviceState(dict):
    """name -> DeviceQuantity (plus 'time' -> datetime).  Built once from a host state for a set of components."""

    def __init__(self, ctx, wild_names, wild_shape):
        super().__init__()
        self.ctx = ctx
        self.wild_names, self.wild_shape = list(wild_names), list(wild_shape)
        self.ncol = int(np.prod(wild_shape)) if wild_shape else 1
        self.scalars = {}          # zero-dimensional quantities stay on the host (name -> float)
        self.host_dims = {}        # name -> dims of the host DataArray the quantity came from (download restores them)
        self._work = {}            # derived arrays and component outputs, by key: allocated once, reused every step
        self._tables = set()
        self.step_count = 0
        self._derived_ok = False   # interface temperatures / vmr / cos(zenith) match the resident state

    def __setitem__(self, name, value):
        self._derived_ok = False
        super().__setitem__(name, value)

    def update(self, *args, **kwargs):
        self._derived_ok = False
        super().update(*args, **kwargs)

    # ---- construction --------------------------------------------------------------------------------------------
    @classmethod
    def from_host(cls, state, components, device=0, context=None, deferred=True):
        ctx = context if context is not None else make_context(device)
        props = {}
        for comp in components:
            comp = getattr(comp, "component", comp)          # UpdateFrequencyWrapper
            for name, prop in comp.input_properties.items():
                props.setdefault(name, prop)
        wild_names = wild_shape = None
        for name, prop in props.items():
            if name in state and "*" in prop["dims"]:
                da = state[name]
                named = [d for d in prop["dims"] if d != "*"]
                wn = [d for d in da.dims if d not in named]
                if wild_names is None or len(wn) > len(wild_names):
                    wild_names, wild_shape = wn, [np.asarray(da.values).shape[da.dims.index(d)] for d in wn]
        self = cls(ctx, wild_names or [], wild_shape or [])
        self["time"] = state.get("time")
        for name, prop in props.items():
            if name not in state:
                continue          # produced by another component during the step (zenith angle, fluxes ...)
            self.put_host(name, state[name], prop)
        if deferred:
            self._prev_deferred = ctx.set_deferred(True)
        return self

    def close(self):
        """Collect the pending work and hand the (shared) context back in the mode it was found in: from_host switches the
        context to deferred mode, which every other memspace=1 caller of the same context would otherwise inherit."""
        prev = getattr(self, "_prev_deferred", None)
        if prev is not None:
            self.ctx.synchronize()
            self.ctx.set_deferred(prev)
            self._prev_deferred = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def put_host(self, name, da, prop):
        """Upload a host DataArray in the layout / units of `prop` (a component's input property)."""
        values, dims = np.asarray(da.values), tuple(da.dims)
        want = list(prop["dims"])
        if name in _SURFACE_ROW:          # always resident as the radiation writes them, whoever asked first
            want = ["interface_levels", "*"]
        self.host_dims[name] = dims
        if values.dtype.kind not in "fiub":      # area_type strings -> codes (slab_surface.py:9)
            from .slab_surface import AREA_MAP
            codes = np.zeros(values.shape, dtype=np.int32)
            s = values.astype(str)
            for k, v in AREA_MAP.items():
                codes[s == k] = v
            self[name] = DeviceQuantity(_hip.DeviceArray.from_host(np.ascontiguousarray(codes.reshape(-1))), (codes.size,), ("*",), prop["units"])
            return
        values = _convert(values.astype(np.float64), _units_of(da), prop["units"])
        if not want:
            self.scalars[name] = float(values)
            return
        named = [d for d in want if d != "*"]
        wild = [d for d in dims if d not in named]
        if wild and sorted(wild) == sorted(self.wild_names):
            wild = self.wild_names
        order = []
        for d in want:
            order.extend([dims.index(w) for w in (wild if d == "*" else [d])])
        shape = [self.ncol if d == "*" else values.shape[dims.index(d)] for d in want]
        arr = np.ascontiguousarray(np.transpose(values, order).reshape(shape))
        self[name] = DeviceQuantity(_hip.DeviceArray.from_host(arr), shape, want, prop["units"])

    # ---- buffers -------------------------------------------------------------------------------------------------
    def work(self, key, shape, dims, units, dtype=np.float64):
        """A device array owned by the state, allocated on first use (component outputs, derived inputs)."""
        q = self._work.get(key)
        if q is None or q.shape != tuple(shape):
            q = self._work[key] = DeviceQuantity(_hip.DeviceArray(shape, dtype), shape, dims, units)
        return q

    def need(self, name, prop):
        q = self.get(name)
        if q is None:
            raise KeyError("device state is missing input quantity %r" % name)
        if tuple(q.dims) != tuple(prop["dims"]):
            raise ValueError("quantity %r is resident as %s, component wants %s" % (name, q.dims, prop["dims"]))
        if q.buf.dtype == np.float64 and not _same_units(q.units, prop["units"]):
            raise ValueError("quantity %r is resident in %r, component wants %r" % (name, q.units, prop["units"]))
        return q

    def init_tables(self, which, cpd):
        (self.ctx.sw_init if which == "sw" else self.ctx.lw_init)(cpd)      # (idempotent per context)

    # ---- back to the host ----------------------------------------------------------------------------------------
    def join_streams(self):
        """Main-stream work enqueued from here on runs after the longwave that is in flight on its own stream.  Called by
        whatever consumes the longwave's outputs or overwrites its inputs on the main stream -- NOT by the shortwave call, so a
        stepper that calls the longwave and then the shortwave gets the two side by side (ordering the main stream behind the
        longwave inside the longwave call, as this did at first, ran them back to back: 1.87 instead of 1.6 ms per step)."""
        if getattr(self, "_lw_inflight", False):
            self.ctx.order_streams(1)
            self._lw_inflight = False

    def download(self, name, synchronize=True):
        """The quantity as a host DataArray: wildcard re-expanded to the horizontal dims it was uploaded with."""
        if synchronize:
            self.ctx.synchronize()
        if name in self.scalars:
            return _sc.DataArray(np.array(self.scalars[name]), dims=(), attrs={"units": ""})
        q = self[name]
        arr = q.buf.download().reshape(q.shape)
        shape, names = [], []
        for d, n in zip(q.dims, q.shape):
            if d == "*":
                shape.extend(self.wild_shape); names.extend(self.wild_names)
            else:
                shape.append(n); names.append(d)
        return _sc.DataArray(arr.reshape(shape), dims=names, attrs={"units": q.units})


# radiation outputs [interface_levels][*] of which the slab reads the surface row in place
_SURFACE_ROW = ("downwelling_shortwave_flux_in_air", "downwelling_longwave_flux_in_air", "upwelling_shortwave_flux_in_air",
                "upwelling_longwave_flux_in_air")


# ---- the components' device paths ---------------------------------------------------------------------------------
def _derived(ds, comp_inputs):
    """Interface temperatures, water-vapour volume mixing ratio and cos(zenith) from the resident state, once per step."""
    if ds._derived_ok:
        return ds._work["derived_vals"]
    ctx = ds.ctx
    ds.join_streams()      # a longwave still in flight reads the buffers rewritten here
    t, p, pi = ds["air_temperature"], ds["air_pressure"], ds["air_pressure_on_interface_levels"]
    nlay, ncol = t.shape
    out = {}
    tint = ds.work("d.tint", (nlay + 1, ncol), ("interface_levels", "*"), "degK")
    ctx.interface_values(ncol, nlay, t.ptr, ds["surface_temperature"].ptr, p.ptr, pi.ptr, tint.ptr)
    out["tint"] = tint
    q = ds.work("d.h2o", (nlay, ncol), ("mid_levels", "*"), "dimensionless")
    ctx.elementwise("muldiv", nlay * ncol, ds["specific_humidity"].ptr, q.ptr, alpha=28.964, beta=18.02)
    out["h2o"] = q
    if "zenith_angle" in ds:
        cz = ds.work("d.coszen", (ncol,), ("*",), "dimensionless")
        ctx.elementwise("cos", ncol, ds["zenith_angle"].ptr, cz.ptr)
        out["coszen"] = cz
    ctx.order_streams(0)          # the longwave stream may start once these (and everything before them) are done
    ds._work["derived_vals"] = out
    ds._derived_ok = True
    return out


def shortwave_device_call(self, ds):
    """RRTMGShortwave on a DeviceState: the body of array_call (sw/component.py:472-668) with device pointers."""
    ds.init_tables("sw", self._Cpd)
    P = self.input_properties
    g = lambda n: ds.need(n, P[n]).ptr
    der = _derived(ds, P)
    nlay, ncol = ds["air_temperature"].shape
    day_of_year = 0 if self._ignore_day_of_year else ds["time"].timetuple().tm_yday
    inp = dict(
        ncol=ncol, nlay=nlay, play=g("air_pressure"), plev=g("air_pressure_on_interface_levels"), tlay=g("air_temperature"), tlev=der["tint"].ptr,
        tsfc=g("surface_temperature"), h2o=der["h2o"].ptr, o3=g("mole_fraction_of_ozone_in_air"), co2=g("mole_fraction_of_carbon_dioxide_in_air"),
        ch4=g("mole_fraction_of_methane_in_air"), n2o=g("mole_fraction_of_nitrous_oxide_in_air"), o2=g("mole_fraction_of_oxygen_in_air"),
        asdir=g("surface_albedo_for_direct_shortwave"), asdif=g("surface_albedo_for_diffuse_shortwave"),
        aldir=g("surface_albedo_for_direct_near_infrared"), aldif=g("surface_albedo_for_diffuse_near_infrared"),
        coszen=der["coszen"].ptr, cldfr=g("cloud_area_fraction_in_atmosphere_layer"),
        taucld=g("shortwave_optical_thickness_due_to_cloud"), ssacld=g("single_scattering_albedo_due_to_cloud"),
        asmcld=g("cloud_asymmetry_parameter"), fsfcld=g("cloud_forward_scattering_fraction"),
        cicewp=g("mass_content_of_cloud_ice_in_atmosphere_layer"), cliqwp=g("mass_content_of_cloud_liquid_water_in_atmosphere_layer"),
        reice=g("cloud_ice_particle_size"), reliq=g("cloud_water_droplet_radius"),
        tauaer=g("shortwave_optical_thickness_due_to_aerosol"), ssaaer=g("single_scattering_albedo_due_to_aerosol"),
        asmaer=g("aerosol_asymmetry_parameter"), ecaer=g("aerosol_optical_depth_at_55_micron"),
        bndsolvar=self._solar_var_by_band, indsolvar=self._fac_sunspot_coeff,
        icld=self._cloud_overlap, iaer=self._aerosol_type, inflg=self._cloud_optics, iceflg=self._ice_props, liqflg=self._liq_props,
        dyofyr=day_of_year, isolvar=self._solar_var_flag, scon=float(self._solar_const),
        adjes=ds.scalars["flux_adjustment_for_earth_sun_distance"], solcycfrac=ds.scalars["solar_cycle_fraction"])
    if self._mcica:
        if self._random_number_generator == 0:
            self._permute_seed = np.random.randint(0, 1024)
        elif self._random_number_generator == 1:
            self._permute_seed = np.random.randint(0, 2 ** 31 - 1)
        inp.update(irng=self._random_number_generator, permuteseed=self._permute_seed)
    il, ml = (nlay + 1, ncol), (nlay, ncol)
    self._device_calls = getattr(self, "_device_calls", 0) + 1          # outputs alternate between two buffer sets: the
    w = lambda key, shape, dims, units: ds.work(("sw", id(self), key, self._device_calls & 1), shape, dims, units)   # state may still hold the last ones
    fl = {k: w(k, il, ("interface_levels", "*"), "W m^-2") for k in ("swuflx", "swdflx", "swuflxc", "swdflxc")}
    hr, hrc = w("swhr", ml, ("mid_levels", "*"), "degK day^-1"), w("swhrc", ml, ("mid_levels", "*"), "degK day^-1")
    out = {k: v.ptr for k, v in fl.items()}
    out.update(swhr=hr.ptr, swhrc=hrc.ptr)
    ds.ctx.sw_fluxes(inp, mcica=self._mcica, out=out, memspace=1)
    diagnostics = {
        "upwelling_shortwave_flux_in_air": fl["swuflx"], "downwelling_shortwave_flux_in_air": fl["swdflx"],
        "upwelling_shortwave_flux_in_air_assuming_clear_sky": fl["swuflxc"], "downwelling_shortwave_flux_in_air_assuming_clear_sky": fl["swdflxc"],
        "air_temperature_tendency_from_shortwave_assuming_clear_sky": hrc, "air_temperature_tendency_from_shortwave": hr}
    return {"air_temperature": hr}, diagnostics


def longwave_device_call(self, ds):
    """RRTMGLongwave on a DeviceState: the body of array_call (lw/component.py:373-522) with device pointers."""
    ds.init_tables("lw", self._Cpd)
    P = self.input_properties
    g = lambda n: ds.need(n, P[n]).ptr
    der = _derived(ds, P)
    nlay, ncol = ds["air_temperature"].shape
    tlev = der["tint"].ptr if self._calc_Tint else g("air_temperature_on_interface_levels")
    inp = dict(
        ncol=ncol, nlay=nlay, play=g("air_pressure"), plev=g("air_pressure_on_interface_levels"), tlay=g("air_temperature"), tlev=tlev,
        tsfc=g("surface_temperature"), h2o=der["h2o"].ptr, o3=g("mole_fraction_of_ozone_in_air"), co2=g("mole_fraction_of_carbon_dioxide_in_air"),
        ch4=g("mole_fraction_of_methane_in_air"), n2o=g("mole_fraction_of_nitrous_oxide_in_air"), o2=g("mole_fraction_of_oxygen_in_air"),
        cfc11=g("mole_fraction_of_cfc11_in_air"), cfc12=g("mole_fraction_of_cfc12_in_air"), cfc22=g("mole_fraction_of_cfc22_in_air"),
        ccl4=g("mole_fraction_of_carbon_tetrachloride_in_air"), emis=g("surface_longwave_emissivity"),
        cldfr=g("cloud_area_fraction_in_atmosphere_layer"), taucld=g("longwave_optical_thickness_due_to_cloud"),
        cicewp=g("mass_content_of_cloud_ice_in_atmosphere_layer"), cliqwp=g("mass_content_of_cloud_liquid_water_in_atmosphere_layer"),
        reice=g("cloud_ice_particle_size"), reliq=g("cloud_water_droplet_radius"), tauaer=g("longwave_optical_thickness_due_to_aerosol"),
        icld=self._cloud_overlap, idrv=self._calc_dflxdt, inflg=self._cloud_optics, iceflg=self._ice_props, liqflg=self._liq_props)
    if self._mcica:
        if self._random_number_generator == 0:
            self._permute_seed = np.random.randint(0, 1024)
        elif self._random_number_generator == 1:
            self._permute_seed = np.random.randint(0, 2 ** 31 - 1)
        inp.update(irng=self._random_number_generator, permuteseed=self._permute_seed)
    il, ml = (nlay + 1, ncol), (nlay, ncol)
    self._device_calls = getattr(self, "_device_calls", 0) + 1
    w = lambda key, shape, dims, units: ds.work(("lw", id(self), key, self._device_calls & 1), shape, dims, units)
    fl = {k: w(k, il, ("interface_levels", "*"), "W m^-2") for k in ("uflx", "dflx", "uflxc", "dflxc")}
    hr, hrc = w("hr", ml, ("mid_levels", "*"), "degK day^-1"), w("hrc", ml, ("mid_levels", "*"), "degK day^-1")
    out = {k: v.ptr for k, v in fl.items()}
    out.update(hr=hr.ptr, hrc=hrc.ptr)
    if self._calc_dflxdt:
        du, duc = w("duflx_dt", il, ("interface_levels", "*"), "W m^-2 K^-1"), w("duflxc_dt", il, ("interface_levels", "*"), "W m^-2 K^-1")
        out.update(duflx_dt=du.ptr, duflxc_dt=duc.ptr)
        self.change_in_upward_flux_with_surface_temperature, self.change_in_clear_sky_upward_flux_with_surface_temperature = du, duc
    ds.ctx.lw_fluxes(inp, mcica=self._mcica, out=out, memspace=1)
    ds._lw_inflight = True      # consumers on the main stream (tendency sum, slab, the next derived fields) join first: join_streams()
    diagnostics = {
        "upwelling_longwave_flux_in_air": fl["uflx"], "downwelling_longwave_flux_in_air": fl["dflx"],
        "upwelling_longwave_flux_in_air_assuming_clear_sky": fl["uflxc"], "downwelling_longwave_flux_in_air_assuming_clear_sky": fl["dflxc"],
        "air_temperature_tendency_from_longwave_assuming_clear_sky": hrc, "air_temperature_tendency_from_longwave": hr}
    return {"air_temperature": hr}, diagnostics


def instellation_device_call(self, ds):
    """Instellation on a DeviceState: latitude / longitude resident, time arithmetic on the host (component.py:64-82)."""
    from .instellation import days_from_2000
    lat, lon = ds.need("latitude", self.input_properties["latitude"]), ds.need("longitude", self.input_properties["longitude"])
    self._device_calls = getattr(self, "_device_calls", 0) + 1
    zen = ds.work(("sun", id(self), self._device_calls & 1), (ds.ncol,), ("*",), "radians")
    ds.ctx.zenith_angle(lat.ptr, lon.ptr, days_from_2000(ds["time"]) / 36525.0, out=zen.ptr, memspace=1, ncol=ds.ncol)
    return {"zenith_angle": zen}


def slab_device_call(self, ds):
    """SlabSurface on a DeviceState: rrtmg_hip_slab_surface reads the SURFACE ROW of the [level][column] radiation outputs in place."""
    names = dict(sw_down="downwelling_shortwave_flux_in_air", lw_down="downwelling_longwave_flux_in_air", sw_up="upwelling_shortwave_flux_in_air",
                 lw_up="upwelling_longwave_flux_in_air", lh="surface_upward_latent_heat_flux", sh="surface_upward_sensible_heat_flux",
                 up_heat_soil="upward_heat_flux_at_ground_level_in_soil", heat_flux_sea_ice="heat_flux_into_sea_water_due_to_sea_ice",
                 sea_water_dens="sea_water_density", surf_dens="surface_material_density", heat_cap_soil="heat_capacity_of_soil",
                 surf_therm_cap="surface_thermal_capacity", ocean_mix_thick="ocean_mixed_layer_thickness", soil_layer_thick="soil_layer_thickness",
                 ocean_heat_transport="ocean_heat_transport_convergence")
    ds.join_streams()      # the longwave fluxes in the state may be this step's
    ptrs = {}
    for k in SLAB_IN:
        q = ds[names[k]]
        if names[k] in _SURFACE_ROW:
            if q.dims[0] != "interface_levels":      # uploaded from the host in the slab's own ['*', 'interface_levels'] layout
                raise ValueError("%s must be resident as [interface_levels][*] (a radiation output) for the in-place surface row" % names[k])
        ptrs[k] = q.ptr                               # row 0 = the surface
    self._device_calls = getattr(self, "_device_calls", 0) + 1
    tend = ds.work(("slab", id(self), "tend", self._device_calls & 1), (ds.ncol,), ("*",), "degK s^-1")
    depth = ds.work(("slab", id(self), "depth", self._device_calls & 1), (ds.ncol,), ("*",), "m")
    ds.ctx.slab_surface_device(ds.ncol, ptrs, ds["area_type"].ptr, tend.ptr, depth.ptr)
    return {"surface_temperature": tend}, {"depth_of_slab_surface": depth, "ocean_heat_transport_convergence": ds["ocean_heat_transport_convergence"]}


# ---- time stepping on the device ------------------------------------------------------------------------------------
_AB = {1: (1.0,), 2: (1.5, -0.5), 3: (23.0 / 12.0, -16.0 / 12.0, 5.0 / 12.0), 4: (55.0 / 24.0, -59.0 / 24.0, 37.0 / 24.0, -9.0 / 24.0)}


def _rate(units):
    u = units.strip()
    for suffix, seconds in ((" s^-1", 1.0), ("/s", 1.0), (" day^-1", 86400.0), ("/day", 86400.0)):
        if u.endswith(suffix):
            return 1.0 / seconds
    raise ValueError("tendency units %r are not a rate" % units)


class DeviceAdamsBashforth:
    """sympl's AdamsBashforth around TendencyComponents (tests/test_components.py:123-160), on a DeviceState: tendencies of
    all components summed in "<state units> per second", order ramping 1 -> 2 -> 3, prognostic quantities replaced in the
    state -- every sum and the update are kernels on the context's main stream.  Returns (diagnostics, the same state).
    wait_every_step=False: the host does not wait for the step (a time loop that reads nothing back keeps the GPU busy while
    the host prepares the next step); the device-side `stop` conditions are sticky and surface at the next download(),
    Context.synchronize() or waiting step."""

    def __init__(self, *components, order=3, wait_every_step=True):
        self._wait = bool(wait_every_step)
        if len(components) == 1 and isinstance(components[0], (list, tuple)):
            components = tuple(components[0])
        if order not in _AB:
            raise ValueError("order must be 1..4")
        self.component_list, self._order = list(components), order
        self._history, self._timestep, self._slot = [], None, 0

    def __call__(self, ds, timestep):
        if not isinstance(timestep, datetime.timedelta):
            raise TypeError("timestep must be a datetime.timedelta")
        if self._timestep is None:
            self._timestep = timestep
        elif timestep != self._timestep:
            raise ValueError("timestep must be constant for Adams-Bashforth time stepping")
        ctx, total, diagnostics = ds.ctx, {}, {}
        slot = self._slot = (self._slot + 1) % (self._order + 1)
        # every component first (the longwave goes to its own stream: the shortwave enqueued behind it on the main stream runs
        # beside it), then the sums in component order
        results = []
        for comp in self.component_list:
            tend, diag = comp(ds)
            overlap = set(diag) & set(diagnostics)
            if overlap:
                raise ValueError("two components compute the same diagnostics: %s" % sorted(overlap))
            diagnostics.update(diag)
            results.append(tend)
        ds.join_streams()
        for tend in results:
            for name, q in tend.items():
                acc = ds.work(("ab", id(self), name, slot), q.shape, q.dims, ds[name].units + " s^-1")
                if name in total:
                    ctx.elementwise("axpby", q.size, acc.ptr, acc.ptr, b=q.ptr, alpha=1.0, beta=_rate(q.units))
                else:
                    ctx.elementwise("axpby", q.size, q.ptr, acc.ptr, alpha=_rate(q.units))
                    total[name] = acc
        self._history = [total] + self._history[: self._order - 1]
        dt = timestep.total_seconds()
        for name in total:
            old = ds[name]
            # the contiguous most-recent run of steps that carry a tendency for `name` sets the order for THAT quantity
            # (the coefficients of an order sum to one; a truncated higher-order set would not)
            hist = []
            for h in self._history:
                if name not in h:
                    break
                hist.append(h[name].ptr)
            new = ds.work(("ab", id(self), name, "state", ds.step_count % 2), old.shape, old.dims, old.units)
            ctx.ab_step(old.size, old.ptr, hist, _AB[len(hist)], dt, new.ptr)
            ds[name] = new
        ds.step_count += 1
        if self._wait:
            ctx.synchronize()      # device-side `stop` conditions of this step surface here
        return diagnostics, ds
