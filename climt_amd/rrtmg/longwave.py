"""RRTMGLongwave -- drop-in for climt.RRTMGLongwave (climt/_components/rrtmg/lw/component.py:30-522)
running on librrtmg_hip.so (MI355X).

NOTE: the reference checkout lacks the longwave k-distribution data file (rrtmg_lw_k_g.f90 is a missing
blob), so the table file shipped with this build carries SYNTHETIC k-tables (Context.lw_tables_synthetic()).
The algorithm is parity-checked against the reference Fortran on those tables, but the fluxes are not
physical.  The component therefore FAILS CLOSED: constructing it on synthetic tables raises, unless the
caller opts in with allow_synthetic_tables=True (or RRTMG_HIP_ALLOW_SYNTHETIC_LW=1) -- the parity tests and
the benchmark do.  With the reference data at hand: `python tools/pack_tables.py lw <dir with rrtmg_lw_k_g.f90>`
writes climt_amd/data/rrtmg_lw_data.bin (or point RRTMG_HIP_LW_DATA at the packed file); no code change."""
import logging
import os

import numpy as np

from .._sympl_compat import TendencyComponent, get_constant
from .._util import ensure_contiguous_state
from .common import (UNIT_FACTOR_ON_DEVICE, InputStaging, library_scales, OutputPool, make_context, output_arrays, rrtmg_cloud_ice_props_dict, rrtmg_cloud_liquid_props_dict, rrtmg_cloud_overlap_method_dict,
                     rrtmg_cloud_props_dict, rrtmg_random_number_dict)


def _prop(dims, units):
    return {"dims": list(dims), "units": units}


_ML, _IL = ["mid_levels", "*"], ["interface_levels", "*"]


class RRTMGLongwave(TendencyComponent):
    """The Rapid Radiative Transfer Model (RRTMG), longwave, on AMD MI355X."""

    num_longwave_bands = 16
    num_reduced_g_intervals = 140
    rrtm_iplon = 1
    _unit_factor_on_device = UNIT_FACTOR_ON_DEVICE   # (see common.library_scales)

    input_properties = {
        "air_pressure": _prop(_ML, "mbar"),
        "air_pressure_on_interface_levels": _prop(_IL, "mbar"),
        "air_temperature": _prop(_ML, "degK"),
        "surface_temperature": _prop(["*"], "degK"),
        "specific_humidity": _prop(_ML, "g/g"),
        "mole_fraction_of_ozone_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_carbon_dioxide_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_methane_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_nitrous_oxide_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_oxygen_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_cfc11_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_cfc12_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_cfc22_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_carbon_tetrachloride_in_air": _prop(_ML, "dimensionless"),
        "surface_longwave_emissivity": _prop(["num_longwave_bands", "*"], "dimensionless"),
        "cloud_area_fraction_in_atmosphere_layer": _prop(_ML, "dimensionless"),
        "longwave_optical_thickness_due_to_cloud": _prop(["mid_levels", "*", "num_longwave_bands"], "dimensionless"),
        "mass_content_of_cloud_ice_in_atmosphere_layer": _prop(_ML, "g m^-2"),
        "mass_content_of_cloud_liquid_water_in_atmosphere_layer": _prop(_ML, "g m^-2"),
        "cloud_ice_particle_size": _prop(_ML, "micrometer"),
        "cloud_water_droplet_radius": _prop(_ML, "micrometer"),
        "longwave_optical_thickness_due_to_aerosol": _prop(["num_longwave_bands", "mid_levels", "*"], "dimensionless"),
    }

    tendency_properties = {"air_temperature": _prop(_ML, "degK day^-1")}

    diagnostic_properties = {
        "upwelling_longwave_flux_in_air": _prop(_IL, "W m^-2"),
        "downwelling_longwave_flux_in_air": _prop(_IL, "W m^-2"),
        "upwelling_longwave_flux_in_air_assuming_clear_sky": _prop(_IL, "W m^-2"),
        "downwelling_longwave_flux_in_air_assuming_clear_sky": _prop(_IL, "W m^-2"),
        "air_temperature_tendency_from_longwave_assuming_clear_sky": _prop(_ML, "degK day^-1"),
        "air_temperature_tendency_from_longwave": _prop(_ML, "degK day^-1"),
    }

    def __init__(self, calculate_change_up_flux=False, cloud_overlap_method=None, cloud_optical_properties="liquid_and_ice_clouds",
                 cloud_ice_properties="ebert_curry_two", cloud_liquid_water_properties="radius_dependent_absorption",
                 calculate_interface_temperature=True, mcica=False, random_number_generator="mersenne_twister", device=0,
                 allow_synthetic_tables=False, **kwargs):
        """Same keyword arguments and defaults as climt.RRTMGLongwave (lw/component.py:167-178); additions: `device`
        (GPU ordinal) and `allow_synthetic_tables` (see the module docstring)."""
        self.input_properties = RRTMGLongwave.input_properties.copy()
        self._calc_dflxdt = 1 if calculate_change_up_flux else 0
        self._mcica = mcica
        if mcica:
            self._permute_seed = None
            self._random_number_generator = rrtmg_random_number_dict[random_number_generator.lower()]
            # messages asserted by the reference's tests (tests/test_components.py:454-461)
            if type(cloud_overlap_method) is str:
                if cloud_overlap_method.lower() == "clear_only":
                    logging.info("cloud_overlap_method == 'clear_only'."
                                 " This overrides all other properties. "
                                 "There are no clouds.")
            if cloud_optical_properties.lower() == "single_cloud_type":
                logging.warning("cloud_optical_properties must be 'direct_input' or "
                                "'liquid_and_ice_clouds' for radiative calculations with "
                                "clouds using McICA.")
        if cloud_overlap_method is None:
            cloud_overlap_method = "random"
        self._cloud_overlap = rrtmg_cloud_overlap_method_dict[cloud_overlap_method.lower()]
        self._cloud_optics = rrtmg_cloud_props_dict[cloud_optical_properties.lower()]
        self._ice_props = rrtmg_cloud_ice_props_dict[cloud_ice_properties.lower()]
        self._liq_props = rrtmg_cloud_liquid_props_dict[cloud_liquid_water_properties.lower()]
        self._calc_Tint = calculate_interface_temperature
        self._Cpd = get_constant("heat_capacity_of_dry_air_at_constant_pressure", "J/kg/K")
        if not self._calc_Tint:
            self.input_properties["air_temperature_on_interface_levels"] = _prop(_IL, "degK")
        self._ctx = make_context(device)
        self._pool = OutputPool()
        self._input_staging = InputStaging()
        self._ctx.lw_init(self._Cpd)
        if self._ctx.lw_tables_synthetic():
            msg = ("RRTMGLongwave: the longwave k-distribution tables in this build are SYNTHETIC (the reference data "
                   "file rrtmg_lw_k_g.f90 was not available); fluxes and heating rates are not physical.")
            if not (allow_synthetic_tables or os.environ.get("RRTMG_HIP_ALLOW_SYNTHETIC_LW", "") not in ("", "0")):
                raise RuntimeError(msg + "  Pack the real tables (tools/pack_tables.py lw) or pass allow_synthetic_tables=True "
                                         "/ set RRTMG_HIP_ALLOW_SYNTHETIC_LW=1 to run on them knowingly.")
            logging.warning(msg)
        # derivative of the upward flux w.r.t. surface temperature (idrv = 1), kept on the instance: the
        # reference never hands these arrays back (its Cython shim would fail with calculate_change_up_flux=True)
        self.change_in_upward_flux_with_surface_temperature = None
        self.change_in_clear_sky_upward_flux_with_surface_temperature = None
        super(RRTMGLongwave, self).__init__(**kwargs)

    def __call__(self, state, *args, **kwargs):
        """A host state goes through sympl's machinery to array_call; a climt_amd.DeviceState (state resident in HBM) takes
        the device path: same quantities, DeviceQuantity handles instead of arrays (climt_amd/device_state.py)."""
        from ..device_state import DeviceState, longwave_device_call
        if isinstance(state, DeviceState):
            return longwave_device_call(self, state)
        return super(RRTMGLongwave, self).__call__(state, *args, **kwargs)

    @ensure_contiguous_state
    def array_call(self, state):
        """Longwave heating tendency and up/down fluxes (all-sky and clear-sky)."""
        # mass_to_volume_mixing_ratio(q, 18.02) = q * 28.964 / 18.02 and the unit factors of the pressures and cloud water paths
        # are applied by the library on the device, after the upload (common.library_scales): no host pass over those arrays
        scales, unit = library_scales(state)
        Q = state["specific_humidity"]
        n_layers, n_columns = state["air_temperature"].shape
        # calculate_interface_temperature: the log-pressure interpolation (lw/component.py:378-384) is done by the library on
        # the device (tlev = None), not by numpy here -- 4 ms of np.log per call at 128 x 64 x 60
        T_interface = None if self._calc_Tint else state["air_temperature_on_interface_levels"]
        # (recycled when the caller has dropped an earlier call's results: the library overwrites every element)
        diagnostics = output_arrays(self._pool, self.diagnostic_properties, state, self.input_properties)
        tendencies = output_arrays(self._pool, self.tendency_properties, state, self.input_properties)
        inp = dict(
            play=unit["air_pressure"], plev=unit["air_pressure_on_interface_levels"], tlay=state["air_temperature"],
            tlev=T_interface, tsfc=state["surface_temperature"], h2o=Q, o3=state["mole_fraction_of_ozone_in_air"],
            co2=state["mole_fraction_of_carbon_dioxide_in_air"], ch4=state["mole_fraction_of_methane_in_air"],
            n2o=state["mole_fraction_of_nitrous_oxide_in_air"], o2=state["mole_fraction_of_oxygen_in_air"],
            cfc11=state["mole_fraction_of_cfc11_in_air"], cfc12=state["mole_fraction_of_cfc12_in_air"],
            cfc22=state["mole_fraction_of_cfc22_in_air"], ccl4=state["mole_fraction_of_carbon_tetrachloride_in_air"],
            emis=state["surface_longwave_emissivity"], cldfr=state["cloud_area_fraction_in_atmosphere_layer"],
            taucld=state["longwave_optical_thickness_due_to_cloud"],
            cicewp=unit["mass_content_of_cloud_ice_in_atmosphere_layer"],
            cliqwp=unit["mass_content_of_cloud_liquid_water_in_atmosphere_layer"],
            reice=state["cloud_ice_particle_size"], reliq=state["cloud_water_droplet_radius"],
            tauaer=state["longwave_optical_thickness_due_to_aerosol"],
            icld=self._cloud_overlap, idrv=self._calc_dflxdt, inflg=self._cloud_optics, iceflg=self._ice_props,
            liqflg=self._liq_props, **scales
        )
        if self._mcica:
            # a fresh seed on every call, drawn exactly as the reference does (lw/component.py:415-424)
            if self._random_number_generator == 0:
                self._permute_seed = np.random.randint(0, 1024)
            elif self._random_number_generator == 1:
                self._permute_seed = np.random.randint(0, 2 ** 31 - 1)
            inp.update(irng=self._random_number_generator, permuteseed=self._permute_seed)
        out = dict(
            uflx=diagnostics["upwelling_longwave_flux_in_air"], dflx=diagnostics["downwelling_longwave_flux_in_air"],
            hr=tendencies["air_temperature"], uflxc=diagnostics["upwelling_longwave_flux_in_air_assuming_clear_sky"],
            dflxc=diagnostics["downwelling_longwave_flux_in_air_assuming_clear_sky"],
            hrc=diagnostics["air_temperature_tendency_from_longwave_assuming_clear_sky"])
        if self._calc_dflxdt:
            # (computed, not returned -- as in the reference, lw/component.py:386-399.  From the liveness-tracked output pool,
            #  like every other result: an array the caller still holds from an earlier call is never written again)
            for key in ("duflx_dt", "duflxc_dt"):
                out[key] = self._pool.zeros_like_fresh(key, (n_layers + 1, n_columns))
        self._input_staging.wait()
        self._ctx.lw_fluxes(inp, mcica=self._mcica, out=out)
        if self._calc_dflxdt:
            self.change_in_upward_flux_with_surface_temperature = out["duflx_dt"]
            self.change_in_clear_sky_upward_flux_with_surface_temperature = out["duflxc_dt"]
        # the reference aliases (not copies) the tendency here (lw/component.py:518-520)
        diagnostics["air_temperature_tendency_from_longwave"] = tendencies["air_temperature"]
        return tendencies, diagnostics
