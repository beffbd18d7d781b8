"""Grid and default-state generator for the radiation path: `get_grid`, `get_default_state`.

Host-side mirror of climt/_core/initialization.py (get_grid :451-573, get_hybrid_sigma_pressure_levels :625-728,
HybridSigmaPressureDiagnosticComponent :574-622, default value table :758-1030, get_default_state :1096-1127,
init_ozone :1130-1143) restricted to what the components of this package read, so that a model script such as
examples/gmd_aquaplanet.py can build its initial state without the reference installed:

    grid  = get_grid(nx=32, ny=16, nz=28)
    state = get_default_state([RRTMGLongwave(allow_synthetic_tables=True), RRTMGShortwave(), SlabSurface()], grid_state=grid)

Everything here is O(grid) numpy run once at start-up -- not part of the per-step hot path, hence no kernel.
The values are pinned by the reference's own golden caches: their `*_stepping-1.cache` files hold the complete
default state of `get_grid(nz=30)` / `get_grid(nx, ny, nz)` (tests/test_initialization.py).
"""
import os
from datetime import datetime

import numpy as np

from . import _sympl_compat as _sc
from ._sympl_compat import DataArray, get_constant

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# name -> (value, units, domain[, dtype]).  domain = "<where>[_horizontal|_interface]" or None for a scalar
# (initialization.py:758-1030; quantities of components outside this package are kept so that a mixed component list,
# e.g. with climt's own dynamics, still initialises).
_DEFAULTS = {}


def _defaults(domain, units, **values):
    for name, value in values.items():
        _DEFAULTS[name] = (value, units, domain)


_defaults("atmosphere", "degK", air_temperature=290.0)
_defaults("atmosphere", "m/s", northward_wind=0.0, eastward_wind=0.0)
_defaults("atmosphere", "s^-1", divergence_of_wind=0.0, atmosphere_relative_vorticity=0.0)
_defaults("atmosphere", "kg/kg", specific_humidity=0.0)
_defaults("atmosphere", "dimensionless",
          mole_fraction_of_carbon_dioxide_in_air=330e-6, mole_fraction_of_methane_in_air=0.0,
          mole_fraction_of_nitrous_oxide_in_air=0.0, mole_fraction_of_oxygen_in_air=0.21,
          mole_fraction_of_nitrogen_in_air=0.78, mole_fraction_of_hydrogen_in_air=500e-9,
          mole_fraction_of_cfc11_in_air=0.0, mole_fraction_of_cfc12_in_air=0.0, mole_fraction_of_cfc22_in_air=0.0,
          mole_fraction_of_carbon_tetrachloride_in_air=0.0, cloud_area_fraction_in_atmosphere_layer=0.0)
_defaults("atmosphere", "kg m^-2", mass_content_of_cloud_ice_in_atmosphere_layer=0.0,
          mass_content_of_cloud_liquid_water_in_atmosphere_layer=0.0)
_defaults("atmosphere", "micrometer", cloud_ice_particle_size=20.0, cloud_water_droplet_radius=10.0)
_defaults("atmosphere_horizontal", "kg m^-2 s^-1", cloud_base_mass_flux=0.0)
_defaults("atmosphere_horizontal", "radians", zenith_angle=0.0)
_defaults("atmosphere_horizontal", "degK", irradiation_temperature=0.0, internal_temperature=0.0)
_defaults("atmosphere_interface", "W m^-2", downwelling_shortwave_flux_in_air=0.0, downwelling_longwave_flux_in_air=0.0,
          upwelling_shortwave_flux_in_air=0.0, upwelling_longwave_flux_in_air=0.0)
_defaults("surface", "kg/kg", surface_specific_humidity=0.0)
_defaults("surface", "degK", surface_temperature=300.0, soil_surface_temperature=300.0)
_defaults("surface", "m^2 s^-2", surface_geopotential=0.0)
_defaults("surface", "J kg^-1 degK^-1", surface_thermal_capacity=4.1813e3)
_defaults("surface", "m", depth_of_slab_surface=50.0, lwe_thickness_of_soil_moisture_content=0)
_defaults("surface", "kg m^-3", surface_material_density=1000.0)
_defaults("surface", "dimensionless",
          surface_albedo_for_direct_shortwave=0.06, surface_albedo_for_diffuse_shortwave=0.06,
          surface_albedo_for_direct_near_infrared=0.06, surface_albedo_for_diffuse_near_infrared=0.06,
          surface_roughness_length=0.0002, surface_drag_coefficient_for_heat_in_air=0.0012,
          surface_drag_coefficient_for_momentum_in_air=0.0012)
_defaults("surface", "W m^-2", surface_upward_sensible_heat_flux=0.0, surface_upward_latent_heat_flux=0.0)
_defaults("surface", "N m^-2", surface_downward_eastward_stress=0.0, surface_downward_northward_stress=0.0)
_defaults("surface", "mm day^-1", convective_precipitation_rate=0.0)
_defaults("surface", "m s^-1", stratiform_precipitation_rate=0.0)
_defaults("soil_interface", "degK", soil_temperature=285.0)
_defaults("soil_interface", "m^3/m^3", soil_liquid_water_content=0.2, soil_ice_content=0.0)
_defaults("land_horizontal", "m", soil_layer_thickness=50.0, deep_soil_moisture_content=0.25)
_defaults("land_horizontal", "W m^-2", upward_heat_flux_at_ground_level_in_soil=0.0)
_defaults("land_horizontal", "J kg^-1 degK^-1", heat_capacity_of_soil=2000.0)
_defaults("land_horizontal", "degK", deep_soil_temperature=285.0)
_defaults("land_horizontal", "m s^-1", runoff_rate=0.0)
_defaults("ocean_horizontal", "kg m^-3", sea_water_density=1.029e3)
_defaults("ocean_horizontal", "degK", sea_surface_temperature=300.0)
_defaults("ocean_horizontal", "m", ocean_mixed_layer_thickness=50.0)
_defaults("ocean_horizontal", "W m^-2", ocean_heat_transport_convergence=0.0)
_defaults("ice_interface", "degK", snow_and_ice_temperature=270.0)
_defaults("ice_horizontal", "W m^-2", heat_flux_into_sea_water_due_to_sea_ice=0.0)
_defaults("ice_horizontal", "m", land_ice_thickness=0.0, sea_ice_thickness=0.0, surface_snow_thickness=0.0)
_defaults(None, "dimensionless", solar_cycle_fraction=0.0, flux_adjustment_for_earth_sun_distance=1.0)
_DEFAULTS["area_type"] = ("sea", "dimensionless", "surface", "S100")
_DEFAULTS["soil_type"] = ("clay", "dimensionless", "land_horizontal", "S100")

_A_COORD = "atmosphere_hybrid_sigma_pressure_a_coordinate_on_interface_levels"
_B_COORD = "atmosphere_hybrid_sigma_pressure_b_coordinate_on_interface_levels"
_VERTICAL = {"atmosphere": (_A_COORD, "mid_levels", "interface_levels"),
             "ice": ("height_on_ice_interface_levels", "ice_mid_levels", "ice_interface_levels"),
             "soil": ("height_on_soil_interface_levels", "soil_mid_levels", "soil_interface_levels")}

# band counts a radiation scheme may override before the state is built (initialization.py:108-136)
_num_bands = {"longwave": None, "shortwave": None}
NUM_ECMWF_AEROSOLS = 6      # sw/component.py: num_ecmwf_aerosols


def set_num_longwave_bands(n):
    _num_bands["longwave"] = int(n)


def set_num_shortwave_bands(n):
    _num_bands["shortwave"] = int(n)


def _nbands(which):
    if _num_bands[which] is not None:
        return _num_bands[which]
    from .rrtmg import RRTMGLongwave, RRTMGShortwave
    return RRTMGLongwave.num_longwave_bands if which == "longwave" else RRTMGShortwave.num_shortwave_bands


def _quantity(values, units, dims):
    return DataArray(values, dims=tuple(dims), attrs={"units": units})


def _set_constant(name, value, units):
    if _sc.HAVE_SYMPL:  # pragma: no cover
        import sympl
        sympl.set_constant(name, value, units)
    else:
        _sc.set_constant(name, value, units)


# -- vertical coordinate -------------------------------------------------------------------------------------------
def get_hybrid_sigma_pressure_levels(num_levels=28, reference_pressure=1e5, model_top_pressure=20,
                                     proportion_isobaric_levels=0.25, proportion_sigma_levels=0.1):
    """a_k [Pa], b_k of the NEWHYB2 hybrid sigma-pressure coordinate of Eckermann (2009, MWR 137) on `num_levels`
    interfaces ordered surface -> top (initialization.py:625-728).

    Interface spacing follows a sine bump in pressure; the top `proportion_isobaric_levels` of the interfaces are pure
    pressure (b = 0), the bottom `proportion_sigma_levels` pure sigma, and in between b = B**r(B) with the exponent
    r blending from 2.2 aloft to r_sigma at the ground through arctan(5 B)/arctan(5) (:730-743)."""
    n = int(num_levels)
    span = reference_pressure - model_top_pressure
    bump = np.sin(np.linspace(0.1, np.pi - 0.1, n - 1))
    bump /= np.sum(bump)
    bump *= span
    p = np.full(n, float(model_top_pressure))
    p[1:] = model_top_pressure + np.cumsum(bump)            # top -> surface
    sigma = (p - model_top_pressure) / span

    n_iso = int(proportion_isobaric_levels * n)
    n_sig = int(proportion_sigma_levels * n)
    s_iso = sigma[n_iso - 1]
    big_b = (sigma - s_iso) / (1 - s_iso)
    r_sigma = 1.0 if n_sig > 0 else 1.35
    expo = 2.2 + (r_sigma - 2.2) * np.arctan(5 * big_b) / np.arctan(5)

    level = np.arange(n)
    bk = np.where(level < n_iso, 0.0, np.where(level < n - n_sig, np.abs(big_b) ** expo, big_b))
    ak = np.where(level < n_iso, p, model_top_pressure + (sigma - bk) * span)
    return {_A_COORD: _quantity(ak[::-1].copy(), "dimensionless", ("interface_levels",)),
            _B_COORD: _quantity(bk[::-1].copy(), "dimensionless", ("interface_levels",))}


def _pressure_levels(ak, bk, ps):
    """Interface pressures a + b (ps - p_top) and the mid-level pressures of the Simmons-Burridge-like mean
    [(p_{k+1}^{kappa+1} - p_k^{kappa+1}) / ((kappa+1) dp)]^{1/kappa} (initialization.py:601-622)."""
    p_top = get_constant("top_of_model_pressure", "Pa")
    kappa = get_constant("gas_constant_of_dry_air", "J kg^-1 K^-1") / get_constant(
        "heat_capacity_of_dry_air_at_constant_pressure", "J kg^-1 K^-1")
    shape = (-1,) + (1,) * ps.ndim
    p_int = ak.reshape(shape) + bk.reshape(shape) * (ps[None] - p_top)
    lo, hi = p_int[:-1], p_int[1:]
    p_mid = ((hi ** (kappa + 1) - lo ** (kappa + 1)) / ((kappa + 1) * (hi - lo))) ** (1.0 / kappa)
    if np.any(np.isnan(p_mid)):
        raise AssertionError("mid-level pressure is not a number")
    return p_mid, p_int


def gaussian_latitudes(n):
    """Gauss-Legendre latitudes (north first) and the cell edges implied by the quadrature weights, degrees
    (initialization.py:442-448)."""
    x, w = np.polynomial.legendre.leggauss(int(n))
    edges = np.concatenate(([-1.0], -1 + np.cumsum(w[:-1]), [1.0]))
    return -np.rad2deg(np.arcsin(x)), -np.rad2deg(np.arcsin(edges))


def get_grid(nx=None, ny=None, nz=28, n_ice_interface_levels=10, n_soil_interface_levels=4, p_surf_in_Pa=None,
             p_toa_in_Pa=None, proportion_sigma_levels=0.1, proportion_isobaric_levels=0.25, x_name="lon", y_name="lat",
             latitude_grid="gaussian"):
    """Grid state: hybrid coordinate, pressures, surface pressure, longitude/latitude, ice and soil interface heights,
    time = 2000-01-01 (initialization.py:451-573; same arguments, same defaults)."""
    if p_surf_in_Pa is None:
        p_surf_in_Pa = get_constant("reference_air_pressure", "Pa")
    if p_toa_in_Pa is None:
        p_toa_in_Pa = get_constant("top_of_model_pressure", "Pa")
    else:
        _set_constant("top_of_model_pressure", p_toa_in_Pa, "Pa")
    nx = 1 if nx is None else int(nx)
    ny = 1 if ny is None else int(ny)
    horiz = (y_name, x_name)

    grid = get_hybrid_sigma_pressure_levels(nz + 1, p_surf_in_Pa, p_toa_in_Pa, proportion_isobaric_levels, proportion_sigma_levels)
    ps = np.ones((ny, nx)) * p_surf_in_Pa
    grid["surface_air_pressure"] = _quantity(ps, "Pa", horiz)
    grid["time"] = datetime(2000, 1, 1)
    p_mid, p_int = _pressure_levels(grid[_A_COORD].values, grid[_B_COORD].values, ps)
    grid["air_pressure"] = _quantity(p_mid, "Pa", ("mid_levels",) + horiz)
    grid["air_pressure_on_interface_levels"] = _quantity(p_int, "Pa", ("interface_levels",) + horiz)

    lon = np.linspace(0.0, 360.0, nx * 2, endpoint=False)[:-1:2]
    grid["longitude"] = _quantity(np.broadcast_to(lon[None, :], (ny, nx)).copy(), "degrees_east", horiz)
    kind = latitude_grid.lower()
    if kind == "regular":
        lat = np.linspace(-90.0, 90.0, ny * 2 + 1, endpoint=True)[1:-1:2]
    elif kind == "gaussian":
        lat = gaussian_latitudes(ny)[0]
    else:
        raise ValueError("latitude_grid can be either regular or gaussian. Other grid types are currently not supported.")
    grid["latitude"] = _quantity(np.broadcast_to(lat[:, None], (ny, nx)).copy(), "degrees_north", horiz)

    if n_ice_interface_levels is not None:
        grid["height_on_ice_interface_levels"] = _quantity(np.zeros(n_ice_interface_levels), "m", ("ice_interface_levels",))
    if n_soil_interface_levels is not None:
        grid["height_on_soil_interface_levels"] = _quantity(np.linspace(0.0, 2.0, n_soil_interface_levels), "m", ("soil_interface_levels",))
    return grid


# -- default values ------------------------------------------------------------------------------------------------
def _domain_shape(grid, domain):
    """(shape, dims) of a quantity living on `domain` (initialization.py:21-104)."""
    if domain is None:
        return (), ()
    where, _, kind = domain.partition("_")
    hshape, hdims = tuple(grid["latitude"].shape), tuple(grid["latitude"].dims)
    if where == "surface" or kind == "horizontal":
        return hshape, hdims
    if where in ("land", "ocean"):
        raise NotImplementedError("3D %s grids are not yet supported" % where)
    coord, mid, interface = _VERTICAL[where]
    nint = grid[coord].shape[0]
    if kind == "interface":
        return (nint,) + hshape, (interface,) + hdims
    return (nint - 1,) + hshape, (mid,) + hdims


def _constant_default(name, entry, grid, interface=False):
    value, units, domain = entry[:3]
    dtype = entry[3] if len(entry) > 3 else np.float64
    shape, dims = _domain_shape(grid, domain + "_interface" if interface else domain)
    return {name: _quantity(np.broadcast_to(np.array(value, dtype=dtype), shape).copy(), units, dims)}


def not_a_knot_spline(x_new, x, y):
    """Cubic spline through (x, y), x ascending, with not-a-knot end conditions, evaluated at x_new; outside the
    table the end cubics continue (what scipy.interpolate.CubicSpline does by default, which the reference's golden
    caches were made with: climt/_core/interpolate.py:1-178).  The full (n x n) system for the knot second
    derivatives is solved densely -- n = 30 for the ozone table."""
    x, y, x_new = (np.asarray(v, dtype=np.float64) for v in (x, y, x_new))
    n = x.size
    if n < 4:
        raise ValueError("not_a_knot_spline needs at least 4 points")
    h = np.diff(x)
    slope = np.diff(y) / h
    mat = np.zeros((n, n))
    rhs = np.zeros(n)
    rows = np.arange(1, n - 1)
    mat[rows, rows - 1] = h[:-1]
    mat[rows, rows] = 2.0 * (h[:-1] + h[1:])
    mat[rows, rows + 1] = h[1:]
    rhs[1:-1] = 6.0 * np.diff(slope)
    mat[0, :3] = (h[1], -(h[0] + h[1]), h[0])               # S''' continuous across x[1]
    mat[-1, -3:] = (h[-1], -(h[-2] + h[-1]), h[-2])         # ... and across x[n-2]
    m = np.linalg.solve(mat, rhs)
    k = np.clip(np.searchsorted(x, x_new, side="right") - 1, 0, n - 2)
    left, right = x_new - x[k], x[k + 1] - x_new
    return ((m[k] * right ** 3 + m[k + 1] * left ** 3) / (6.0 * h[k])
            + (y[k] / h[k] - m[k] * h[k] / 6.0) * right + (y[k + 1] / h[k] - m[k + 1] * h[k] / 6.0) * left)


def _ozone(grid):
    tab = np.load(os.path.join(_DATA, "ozone_profile.npz"))
    p = grid["air_pressure"]
    return {"mole_fraction_of_ozone_in_air": _quantity(not_a_knot_spline(p.values, tab["pressure_Pa"], tab["mole_fraction"]), "mole/mole", p.dims)}


def _gray_longwave_depth(grid):
    p, ps = grid["air_pressure_on_interface_levels"], grid["surface_air_pressure"]
    return {"longwave_optical_depth_on_interface_levels": _quantity(1.0 * (1.0 - p.values / ps.values[None]), "dimensionless", p.dims)}


def _longwave_band_defaults(grid):
    """initialization.py:139-170: black surface, no cloud / aerosol optical thickness, per band."""
    nb = _nbands("longwave")
    (nz,), h = grid["air_pressure"].shape[:1], tuple(grid["latitude"].shape)
    hd = tuple(grid["latitude"].dims)
    return {"surface_longwave_emissivity": _quantity(np.ones((nb,) + h), "dimensionless", ("num_longwave_bands",) + hd),
            "longwave_optical_thickness_due_to_cloud": _quantity(np.zeros((nz,) + h + (nb,)), "dimensionless", ("mid_levels",) + hd + ("num_longwave_bands",)),
            "longwave_optical_thickness_due_to_aerosol": _quantity(np.zeros((nb, nz) + h), "dimensionless", ("num_longwave_bands", "mid_levels") + hd)}


def _shortwave_band_defaults(grid):
    """initialization.py:173-233."""
    nb = _nbands("shortwave")
    (nz,), h = grid["air_pressure"].shape[:1], tuple(grid["latitude"].shape)
    hd = tuple(grid["latitude"].dims)
    out = {}
    for name, value in (("shortwave_optical_thickness_due_to_cloud", 0.0), ("cloud_asymmetry_parameter", 0.85),
                        ("cloud_forward_scattering_fraction", 0.8), ("single_scattering_albedo_due_to_cloud", 0.9)):
        out[name] = _quantity(np.full((nz,) + h + (nb,), value), "dimensionless", ("mid_levels",) + hd + ("num_shortwave_bands",))
    for name, value in (("shortwave_optical_thickness_due_to_aerosol", 0.0), ("aerosol_asymmetry_parameter", 0.0),
                        ("single_scattering_albedo_due_to_aerosol", 0.5)):
        out[name] = _quantity(np.full((nb, nz) + h, value), "dimensionless", ("num_shortwave_bands", "mid_levels") + hd)
    out["aerosol_optical_depth_at_55_micron"] = _quantity(np.zeros((NUM_ECMWF_AEROSOLS, nz) + h), "dimensionless",
                                                          ("num_ecmwf_aerosols", "mid_levels") + hd)
    return out


_LW_BAND_NAMES = ("surface_longwave_emissivity", "longwave_optical_thickness_due_to_cloud", "longwave_optical_thickness_due_to_aerosol")
_SW_BAND_NAMES = ("shortwave_optical_thickness_due_to_cloud", "cloud_asymmetry_parameter", "cloud_forward_scattering_fraction",
                  "single_scattering_albedo_due_to_cloud", "shortwave_optical_thickness_due_to_aerosol", "aerosol_asymmetry_parameter",
                  "single_scattering_albedo_due_to_aerosol", "aerosol_optical_depth_at_55_micron")
_COMPUTED = {"longwave_optical_depth_on_interface_levels": _gray_longwave_depth, "mole_fraction_of_ozone_in_air": _ozone}
_COMPUTED.update({n: _longwave_band_defaults for n in _LW_BAND_NAMES})
_COMPUTED.update({n: _shortwave_band_defaults for n in _SW_BAND_NAMES})
_SUFFIX = "_on_interface_levels"


def _initialise(name, grid):
    """One missing input -> the quantities its initialiser provides (initialization.py:1045-1078; like there, a band
    default brings its sibling arrays with it)."""
    if name in _DEFAULTS:
        return _constant_default(name, _DEFAULTS[name], grid)
    if name.endswith(_SUFFIX) and name[:-len(_SUFFIX)] in _DEFAULTS:
        return _constant_default(name, _DEFAULTS[name[:-len(_SUFFIX)]], grid, interface=True)
    if name in _COMPUTED:
        return _COMPUTED[name](grid)
    raise NotImplementedError("No initialization method for quantity name {}".format(name))


def aggregate_input_properties(component_list):
    """Union of the components' input_properties (sympl.combine_component_properties when sympl is there; otherwise
    the first listing of a name wins after checking that the named dims agree)."""
    if _sc.HAVE_SYMPL:  # pragma: no cover
        from sympl import combine_component_properties
        return combine_component_properties(component_list, "input_properties")
    merged = {}
    for comp in component_list:
        for name, prop in comp.input_properties.items():
            if name in merged:
                a = sorted(d for d in merged[name]["dims"] if d != "*")
                b = sorted(d for d in prop["dims"] if d != "*")
                both_wild = "*" in merged[name]["dims"] and "*" in prop["dims"]
                if a != b and not both_wild:
                    raise ValueError("components disagree on the dimensions of %r: %s vs %s" % (name, merged[name]["dims"], prop["dims"]))
            else:
                merged[name] = dict(prop)
    return merged


def get_default_state(component_list, grid_state=None, n_ice_interface_levels=30, n_soil_interface_levels=4):
    """A reasonable initial state for `component_list`: the grid quantities plus a default for every input the grid
    does not already hold (initialization.py:1096-1127)."""
    grid = grid_state or get_grid(n_ice_interface_levels=n_ice_interface_levels, n_soil_interface_levels=n_soil_interface_levels)
    state = dict(grid)
    for name in aggregate_input_properties(component_list):
        if name not in grid:
            state.update(_initialise(name, grid))
    return state
