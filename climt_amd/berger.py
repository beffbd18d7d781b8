"""BergerSolarInsolation -- drop-in for climt.BergerSolarInsolation (climt/_components/berger_solar_insolation.py:495-576):
solar insolation and zenith angle from the Berger (1978) spectral solutions of the orbital parameters, as CAM 3 does.

Host (numpy, as in the reference): the orbital series of a model year (`_get_orbital_parameters_functional`, :579-625,
cached per year like the reference's `_orbital_parameters`), the true longitude / Earth-sun distance / declination of
the call's time (:651-668) and the time helpers (:683-693).  Device: the per-column hour angle, zenith angle and
insolation (:671-676) through rrtmg_hip_solar_insolation (include/rrtmg_hip.h).  The coefficient tables are data:
climt_amd/data/berger_tables.npz (tools/pack_berger.py)."""
import os

import numpy as np

from ._sympl_compat import DiagnosticComponent, get_constant
from .rrtmg.common import make_context

_TABLES = None
arcsec_to_degree = 1.0 / 3600.0


def _tables():
    global _TABLES
    if _TABLES is None:
        _TABLES = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "berger_tables.npz")))
    return _TABLES


def get_orbital_parameters(years_since_jan_1_1950):
    """(lambda_m0, eccentricity, omega_tilde, obliquity) -- Berger 1978 eq. 1-6 and the bullets of p. 2365."""
    T, t = _tables(), years_since_jan_1_1950
    obliquity = 23.320556
    obliquity += np.sum(T["A"] * arcsec_to_degree * np.cos((T["f"] * arcsec_to_degree * t + T["delta"]) * np.pi / 180.0))
    obliquity = obliquity * np.pi / 180.0
    # zeta is in radians (CAM 3.0 shr_orb_mod.f90): no degree conversion here
    cos_sum = np.sum(T["P"] * np.cos(T["alpha"] * arcsec_to_degree * t + T["zeta"]))
    sin_sum = np.sum(T["P"] * np.sin(T["alpha"] * arcsec_to_degree * t + T["zeta"]))
    eccentricity_squared = cos_sum * cos_sum + sin_sum * sin_sum
    eccentricity = np.sqrt(eccentricity_squared)
    eccentricity_cubed = eccentricity * eccentricity_squared
    pi_val = np.arctan2(sin_sum, cos_sum)
    if pi_val < 0:
        pi_val += 2.0 * np.pi
    omega_tilde = pi_val * 180.0 / np.pi + 50.439273 * arcsec_to_degree * t + 3.392506
    omega_tilde += np.sum(T["F"] * np.sin((T["f_prime"] * arcsec_to_degree * t + T["delta_prime"]) * np.pi / 180.0))
    omega_tilde = omega_tilde % 360.0
    omega_tilde = omega_tilde * np.pi / 180.0
    beta = np.sqrt(1.0 - eccentricity_squared)
    lambda_m0 = 2.0 * ((0.5 * eccentricity + 0.125 * eccentricity_cubed) * (1.0 + beta) * np.sin(omega_tilde + np.pi)
                       - 0.25 * eccentricity_squared * (0.5 + beta) * np.sin(2 * (omega_tilde + np.pi))
                       + 0.125 * eccentricity_cubed * (1.0 / 3.0 + beta) * np.sin(3 * (omega_tilde + np.pi)))
    return lambda_m0, eccentricity, omega_tilde, obliquity


def years_since_vernal_equinox(dt):
    """Fractional years since last March 20, noon UTC (assumed time of vernal equinox)."""
    year_start = type(dt)(dt.year, 3, 20, 12)
    year_end = type(dt)(dt.year + 1, 3, 20, 12)
    return (dt - year_start).total_seconds() / (year_end - year_start).total_seconds()


def fractional_day(dt):
    day_start = type(dt)(dt.year, dt.month, dt.day)
    return (dt - day_start).total_seconds() / (24.0 * 60.0 * 60.0)


class BergerSolarInsolation(DiagnosticComponent):
    """Determines solar insolation using spectral solutions for orbital constants from Berger 1978, on AMD MI355X."""

    input_properties = {
        "longitude": {"dims": ["*"], "units": "degrees_east"},
        "latitude": {"dims": ["*"], "units": "degrees_north"},
    }

    diagnostic_properties = {
        "solar_insolation": {"dims": ["*"], "units": "W m^-2"},
        "solar_zenith_angle": {"dims": ["*"], "units": "radians"},
        "obliquity": {"dims": [], "units": "radians"},
        "eccentricity": {"dims": [], "units": "radians"},
        "normalized_earth_sun_distance": {"dims": [], "units": "dimensionless"},
    }

    def __init__(self, device=0, context=None, **kwargs):
        self._orbital_parameters = {}
        super(BergerSolarInsolation, self).__init__(**kwargs)
        self._ctx = context if context is not None else make_context(device)

    def array_call(self, state):
        solar_constant = get_constant("stellar_irradiance", "W/m^2")
        lat, lon = state["latitude"], state["longitude"]
        lat_flat, lon_flat = np.reshape(lat, (-1,)), np.reshape(lon, (-1,))
        solar_insolation, solar_zenith_angle, obliquity, eccentricity, rho = self._driver(state["time"], lat_flat, lon_flat, solar_constant)
        return {
            "solar_insolation": np.reshape(solar_insolation, np.shape(lat)),
            "solar_zenith_angle": np.reshape(solar_zenith_angle, np.shape(lat)),
            "obliquity": obliquity,
            "eccentricity": eccentricity,
            "normalized_earth_sun_distance": rho,
        }

    def _driver(self, time, lat, lon, solar_constant):
        year = time.year
        if year not in self._orbital_parameters:
            self._orbital_parameters[year] = get_orbital_parameters(float(year - 1950))
        lambda_m0, eccentricity, omega_tilde, obliquity = self._orbital_parameters[year]
        # scalar part of _get_solar_parameters_np (:651-668)
        eccentricity_squared = eccentricity * eccentricity
        lambda_m = lambda_m0 + years_since_vernal_equinox(time) * 2.0 * np.pi
        temp = lambda_m - (omega_tilde + np.pi)
        sin_temp = np.sin(temp)
        lmbda = lambda_m + eccentricity * (2.0 * sin_temp + eccentricity * (1.25 * np.sin(2 * temp)
                                           + eccentricity * ((13.0 / 12.0) * np.sin(3 * temp) - 0.25 * sin_temp)))
        inverse_rho = (1 + eccentricity * np.cos(lmbda - (omega_tilde + np.pi))) / (1 - eccentricity_squared)
        rho = 1.0 / inverse_rho
        inverse_rho_squared = inverse_rho * inverse_rho
        solar_declination_angle = np.arcsin(np.sin(obliquity) * np.sin(lmbda))
        zen, ins = self._ctx.solar_insolation(lat, lon, np.sin(solar_declination_angle), np.cos(solar_declination_angle),
                                              fractional_day(time), solar_constant * inverse_rho_squared)
        return ins, zen, obliquity, eccentricity, rho
