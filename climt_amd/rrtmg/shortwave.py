"""RRTMGShortwave -- drop-in for climt.RRTMGShortwave (climt/_components/rrtmg/sw/component.py:32-668)
running on librrtmg_hip.so (MI355X).  Same class attributes, keyword options and defaults, property
dictionaries, log messages and (tendencies, diagnostics) contract; the Cython/Fortran calls are replaced
by one rrtmg_hip_sw_fluxes call (include/rrtmg_hip.h)."""
import logging

import numpy as np

from .._sympl_compat import TendencyComponent, get_constant
from .._util import ensure_contiguous_state
from .common import (UNIT_FACTOR_ON_DEVICE, InputStaging, library_scales, OutputPool, make_context, output_arrays, rrtmg_aerosol_input_dict, rrtmg_cloud_ice_props_dict, rrtmg_cloud_liquid_props_dict,
                     rrtmg_cloud_overlap_method_dict, rrtmg_cloud_props_dict, rrtmg_random_number_dict)


def _prop(dims, units):
    return {"dims": list(dims), "units": units}


_ML, _IL, _COL = ["mid_levels", "*"], ["interface_levels", "*"], ["*"]
_CLD = ["mid_levels", "*", "num_shortwave_bands"]
_AER = ["num_shortwave_bands", "mid_levels", "*"]


class RRTMGShortwave(TendencyComponent):
    """The Rapid Radiative Transfer Model (RRTMG), shortwave, on AMD MI355X."""

    num_shortwave_bands = 14
    _unit_factor_on_device = UNIT_FACTOR_ON_DEVICE   # (see common.library_scales)
    num_ecmwf_aerosols = 6
    num_reduced_g_intervals = 112
    rrtm_iplon = 1

    input_properties = {
        "air_pressure": _prop(_ML, "mbar"),
        "air_pressure_on_interface_levels": _prop(_IL, "mbar"),
        "air_temperature": _prop(_ML, "degK"),
        "specific_humidity": _prop(_ML, "dimensionless"),
        "mole_fraction_of_ozone_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_carbon_dioxide_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_methane_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_nitrous_oxide_in_air": _prop(_ML, "dimensionless"),
        "mole_fraction_of_oxygen_in_air": _prop(_ML, "dimensionless"),
        "mass_content_of_cloud_ice_in_atmosphere_layer": _prop(_ML, "g m^-2"),
        "mass_content_of_cloud_liquid_water_in_atmosphere_layer": _prop(_ML, "g m^-2"),
        "cloud_ice_particle_size": _prop(_ML, "micrometer"),
        "cloud_water_droplet_radius": _prop(_ML, "micrometer"),
        "cloud_area_fraction_in_atmosphere_layer": _prop(_ML, "dimensionless"),
        "surface_temperature": _prop(_COL, "degK"),
        "zenith_angle": _prop(_COL, "radians"),
        "surface_albedo_for_direct_shortwave": _prop(_COL, "dimensionless"),
        "surface_albedo_for_direct_near_infrared": _prop(_COL, "dimensionless"),
        "surface_albedo_for_diffuse_near_infrared": _prop(_COL, "dimensionless"),
        "surface_albedo_for_diffuse_shortwave": _prop(_COL, "dimensionless"),
        "shortwave_optical_thickness_due_to_cloud": _prop(_CLD, "dimensionless"),
        "shortwave_optical_thickness_due_to_aerosol": _prop(_AER, "dimensionless"),
        "single_scattering_albedo_due_to_cloud": _prop(_CLD, "dimensionless"),
        "single_scattering_albedo_due_to_aerosol": _prop(_AER, "dimensionless"),
        "cloud_asymmetry_parameter": _prop(_CLD, "dimensionless"),
        "aerosol_asymmetry_parameter": _prop(_AER, "dimensionless"),
        "cloud_forward_scattering_fraction": _prop(_CLD, "dimensionless"),
        "aerosol_optical_depth_at_55_micron": _prop(["num_ecmwf_aerosols", "mid_levels", "*"], "dimensionless"),
        "solar_cycle_fraction": _prop([], "dimensionless"),
        "flux_adjustment_for_earth_sun_distance": _prop([], "dimensionless"),
    }

    # no "dims" here, exactly as the reference (sw/component.py:148-150)
    tendency_properties = {"air_temperature": {"units": "degK day^-1"}}

    diagnostic_properties = {
        "upwelling_shortwave_flux_in_air": _prop(_IL, "W m^-2"),
        "downwelling_shortwave_flux_in_air": _prop(_IL, "W m^-2"),
        "upwelling_shortwave_flux_in_air_assuming_clear_sky": _prop(_IL, "W m^-2"),
        "downwelling_shortwave_flux_in_air_assuming_clear_sky": _prop(_IL, "W m^-2"),
        "air_temperature_tendency_from_shortwave_assuming_clear_sky": _prop(_ML, "degK day^-1"),
        "air_temperature_tendency_from_shortwave": _prop(_ML, "degK day^-1"),
    }

    def __init__(self, cloud_overlap_method=None, cloud_optical_properties="liquid_and_ice_clouds",
                 cloud_ice_properties="ebert_curry_two", cloud_liquid_water_properties="radius_dependent_absorption",
                 solar_variability_method=0, use_solar_constant_from_fortran=False, ignore_day_of_year=False,
                 facular_sunspot_amplitude=None, solar_variability_by_band=None, aerosol_type="no_aerosol", mcica=False,
                 random_number_generator="mersenne_twister", device=0, **kwargs):
        """Same keyword arguments and defaults as climt.RRTMGShortwave (sw/component.py:179-194); `device`
        (GPU ordinal) is the one addition."""
        self._mcica = mcica
        if mcica:
            self._permute_seed = None
            self._random_number_generator = rrtmg_random_number_dict[random_number_generator.lower()]
            # messages asserted by the reference's tests (tests/test_components.py:507-530)
            if type(cloud_overlap_method) is str:
                if cloud_overlap_method.lower() == "clear_only":
                    logging.info("cloud_overlap_method == 'clear_only'."
                                 " This overrides all other properties. "
                                 "There are no clouds.")
            if cloud_optical_properties.lower() == "single_cloud_type":
                logging.warning("cloud_optical_properties must be 'direct_input' or "
                                "'liquid_and_ice_clouds' for radiative calculations with "
                                "clouds using McICA.")
            if cloud_optical_properties.lower() == "liquid_and_ice_clouds":
                if cloud_ice_properties.lower() == "ebert_curry_one":
                    logging.warning("cloud_ice_properties should not be set to "
                                    "'ebert_curry_one' for shortwave calculations with "
                                    "McICA.")
                if cloud_liquid_water_properties.lower() == "radius_independent_absorption":
                    logging.warning("cloud_liquid_water_properties must be set to "
                                    "'radius_dependent_absorption' for use with McICA in "
                                    "the shortwave.")
        if cloud_overlap_method is None:
            cloud_overlap_method = "random"
        self._cloud_overlap = rrtmg_cloud_overlap_method_dict[cloud_overlap_method.lower()]
        self._cloud_optics = rrtmg_cloud_props_dict[cloud_optical_properties.lower()]
        self._ice_props = rrtmg_cloud_ice_props_dict[cloud_ice_properties.lower()]
        self._liq_props = rrtmg_cloud_liquid_props_dict[cloud_liquid_water_properties.lower()]
        self._solar_var_flag = solar_variability_method
        self._ignore_day_of_year = ignore_day_of_year
        self._fac_sunspot_coeff = np.ones(2) if facular_sunspot_amplitude is None else np.asarray(facular_sunspot_amplitude, dtype=float)
        self._solar_var_by_band = np.ones(16) if solar_variability_by_band is None else np.asarray(solar_variability_by_band, dtype=float)
        self._aerosol_type = rrtmg_aerosol_input_dict[aerosol_type.lower()]
        self._solar_const = 0 if use_solar_constant_from_fortran else get_constant("stellar_irradiance", "W/m^2")
        self._Cpd = get_constant("heat_capacity_of_dry_air_at_constant_pressure", "J/kg/K")
        self._ctx = make_context(device)
        self._pool = OutputPool()
        self._input_staging = InputStaging()
        # the reference re-runs rrtmg_sw_ini on every McICA call (sw/component.py:547-560); the tables do
        # not depend on the call, so they are built once here
        self._ctx.sw_init(self._Cpd)
        super(RRTMGShortwave, self).__init__(**kwargs)

    def __call__(self, state, *args, **kwargs):
        """A host state goes through sympl's machinery to array_call; a climt_amd.DeviceState (state resident in HBM) takes
        the device path: same quantities, DeviceQuantity handles instead of arrays (climt_amd/device_state.py)."""
        from ..device_state import DeviceState, shortwave_device_call
        if isinstance(state, DeviceState):
            return shortwave_device_call(self, state)
        return super(RRTMGShortwave, self).__call__(state, *args, **kwargs)

    @ensure_contiguous_state
    def array_call(self, state):
        """Shortwave heating tendency and up/down fluxes (all-sky and clear-sky)."""
        # mass_to_volume_mixing_ratio(q, 18.02) = q * 28.964 / 18.02 and the unit factors of the pressures and cloud water paths
        # are applied by the library on the device, after the upload (common.library_scales): no host pass over those arrays
        scales, unit = library_scales(state)
        Q = state["specific_humidity"]
        assert unit["air_pressure"].shape[0] + 1 == unit["air_pressure_on_interface_levels"].shape[0]
        # (the reference also interpolates interface temperatures here, sw/component.py:492-496; RRTMG_SW never reads them)
        Tint = None
        # (recycled when the caller has dropped an earlier call's results: the library overwrites every element)
        diagnostics = output_arrays(self._pool, self.diagnostic_properties, state, self.input_properties)
        tendencies = output_arrays(self._pool, self.tendency_properties, state, self.input_properties)
        day_of_year = 0 if self._ignore_day_of_year else state["time"].timetuple().tm_yday
        inp = dict(
            play=unit["air_pressure"], plev=unit["air_pressure_on_interface_levels"], tlay=state["air_temperature"], tlev=Tint,
            tsfc=state["surface_temperature"], h2o=Q, o3=state["mole_fraction_of_ozone_in_air"],
            co2=state["mole_fraction_of_carbon_dioxide_in_air"], ch4=state["mole_fraction_of_methane_in_air"],
            n2o=state["mole_fraction_of_nitrous_oxide_in_air"], o2=state["mole_fraction_of_oxygen_in_air"],
            asdir=state["surface_albedo_for_direct_shortwave"], asdif=state["surface_albedo_for_diffuse_shortwave"],
            aldir=state["surface_albedo_for_direct_near_infrared"], aldif=state["surface_albedo_for_diffuse_near_infrared"],
            coszen=np.cos(state["zenith_angle"]), cldfr=state["cloud_area_fraction_in_atmosphere_layer"],
            taucld=state["shortwave_optical_thickness_due_to_cloud"], ssacld=state["single_scattering_albedo_due_to_cloud"],
            asmcld=state["cloud_asymmetry_parameter"], fsfcld=state["cloud_forward_scattering_fraction"],
            cicewp=unit["mass_content_of_cloud_ice_in_atmosphere_layer"],
            cliqwp=unit["mass_content_of_cloud_liquid_water_in_atmosphere_layer"],
            reice=state["cloud_ice_particle_size"], reliq=state["cloud_water_droplet_radius"],
            tauaer=state["shortwave_optical_thickness_due_to_aerosol"], ssaaer=state["single_scattering_albedo_due_to_aerosol"],
            asmaer=state["aerosol_asymmetry_parameter"], ecaer=state["aerosol_optical_depth_at_55_micron"],
            bndsolvar=self._solar_var_by_band, indsolvar=self._fac_sunspot_coeff,
            icld=self._cloud_overlap, iaer=self._aerosol_type, inflg=self._cloud_optics, iceflg=self._ice_props,
            liqflg=self._liq_props, dyofyr=day_of_year, isolvar=self._solar_var_flag, scon=float(self._solar_const),
            adjes=state["flux_adjustment_for_earth_sun_distance"].item(), solcycfrac=state["solar_cycle_fraction"].item(), **scales
        )
        if self._mcica:
            # a fresh seed on every call, drawn exactly as the reference does (sw/component.py:537-545)
            if self._random_number_generator == 0:
                self._permute_seed = np.random.randint(0, 1024)
            elif self._random_number_generator == 1:
                self._permute_seed = np.random.randint(0, 2 ** 31 - 1)
            inp.update(irng=self._random_number_generator, permuteseed=self._permute_seed)
        out = dict(
            swuflx=diagnostics["upwelling_shortwave_flux_in_air"], swdflx=diagnostics["downwelling_shortwave_flux_in_air"],
            swhr=tendencies["air_temperature"], swuflxc=diagnostics["upwelling_shortwave_flux_in_air_assuming_clear_sky"],
            swdflxc=diagnostics["downwelling_shortwave_flux_in_air_assuming_clear_sky"],
            swhrc=diagnostics["air_temperature_tendency_from_shortwave_assuming_clear_sky"])
        self._input_staging.wait()
        self._ctx.sw_fluxes(inp, mcica=self._mcica, out=out)
        diagnostics["air_temperature_tendency_from_shortwave"][:] = tendencies["air_temperature"]
        return tendencies, diagnostics
