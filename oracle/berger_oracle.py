"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of climt's `BergerSolarInsolation` (solar insolation and zenith
angle from Berger-1978 orbital series), a producer upstream of the shortwave.  Never imported by the
product; only tests, smoke() and bench.py's cpu_baseline may use anything under oracle/.

Follows /root/reference/climt/_components/berger_solar_insolation.py:
  orbital parameters (Berger eq. 1-6, bullets p. 2365)  :579-625      vernal-equinox perihelion longitude :628-632
  per-column insolation / zenith kernel                  :636-680      time helpers                        :683-693
The coefficient tables are data (climt_amd/data/berger_tables.npz, packed from :7-490 by tools/pack_berger.py).
Quirk kept: the kernel takes sin/cos of the latitude VALUE given in degrees (:673), as the reference does.
Pinned: tests/test_oracle.py checks it against TestBergerSolarInsolation-{column,3d}-0.cache (1e-8, the reference's criterion).
"""
import os

import numpy as np

_T = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "climt_amd", "data", "berger_tables.npz"))
ARCSEC = 1.0 / 3600.0


def orbital_parameters(years_since_jan_1_1950):
    t = years_since_jan_1_1950
    obliquity = 23.320556
    obliquity += np.sum(_T["A"] * ARCSEC * np.cos((_T["f"] * ARCSEC * t + _T["delta"]) * np.pi / 180.0))
    obliquity = obliquity * np.pi / 180.0
    cos_sum = np.sum(_T["P"] * np.cos(_T["alpha"] * ARCSEC * t + _T["zeta"]))
    sin_sum = np.sum(_T["P"] * np.sin(_T["alpha"] * ARCSEC * t + _T["zeta"]))
    e2 = cos_sum * cos_sum + sin_sum * sin_sum
    e = np.sqrt(e2)
    e3 = e * e2
    pi_val = np.arctan2(sin_sum, cos_sum)
    if pi_val < 0:
        pi_val += 2.0 * np.pi
    omega = pi_val * 180.0 / np.pi + 50.439273 * ARCSEC * t + 3.392506
    omega += np.sum(_T["F"] * np.sin((_T["f_prime"] * ARCSEC * t + _T["delta_prime"]) * np.pi / 180.0))
    omega = (omega % 360.0) * np.pi / 180.0
    beta = np.sqrt(1.0 - e2)
    lambda_m0 = 2.0 * ((0.5 * e + 0.125 * e3) * (1.0 + beta) * np.sin(omega + np.pi)
                       - 0.25 * e2 * (0.5 + beta) * np.sin(2 * (omega + np.pi))
                       + 0.125 * e3 * (1.0 / 3.0 + beta) * np.sin(3 * (omega + np.pi)))
    return lambda_m0, e, omega, obliquity


def years_since_vernal_equinox(dt):
    a, b = type(dt)(dt.year, 3, 20, 12), type(dt)(dt.year + 1, 3, 20, 12)
    return (dt - a).total_seconds() / (b - a).total_seconds()


def fractional_day(dt):
    return (dt - type(dt)(dt.year, dt.month, dt.day)).total_seconds() / (24.0 * 60.0 * 60.0)


def solar_parameters(lat, lon, model_time, solar_constant):
    """-> (solar_insolation, solar_zenith_angle, obliquity, eccentricity, normalized_earth_sun_distance)."""
    lat, lon = np.asarray(lat, dtype=np.float64), np.asarray(lon, dtype=np.float64)
    lambda_m0, e, omega, obliquity = orbital_parameters(float(model_time.year - 1950))
    lambda_m = lambda_m0 + years_since_vernal_equinox(model_time) * 2.0 * np.pi
    temp = lambda_m - (omega + np.pi)
    st = np.sin(temp)
    lmbda = lambda_m + e * (2.0 * st + e * (1.25 * np.sin(2 * temp) + e * ((13.0 / 12.0) * np.sin(3 * temp) - 0.25 * st)))
    inverse_rho = (1 + e * np.cos(lmbda - (omega + np.pi))) / (1 - e * e)
    delta = np.arcsin(np.sin(obliquity) * np.sin(lmbda))
    H = 2 * np.pi * (fractional_day(model_time) + lon / 360.0)
    cos_mu = np.sin(lat) * np.sin(delta) - np.cos(lat) * np.cos(delta) * np.cos(H)
    return solar_constant * inverse_rho * inverse_rho * cos_mu, np.arccos(cos_mu), obliquity, e, 1.0 / inverse_rho
