"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of climt's `Instellation` zenith-angle path, the producer of the
`zenith_angle` the shortwave consumes.  Never imported by the product (climt_amd/); only tests,
smoke() and bench.py's cpu_baseline may use anything under oracle/.

Follows /root/reference/climt/_components/instellation/component.py:
  days_from_2000 / total_days     :64-76     zenith kernel            :85-135
  obliquity                       :138-152   ecliptic longitude       :155-179     Greenwich sidereal time :182-191
and, for reproducing the inputs of the reference's golden caches (TestInstellation-{column,3d}-0.cache),
climt/_core/initialization.py: gaussian_latitudes :442-448, longitudes :520-524, time :518.
Pinned: tests/test_oracle.py checks it against both caches to the reference's own 1e-8 criterion.
"""
import datetime

import numpy as np
from numpy.polynomial.legendre import leggauss


def days_from_2000(model_time):
    dt = model_time - datetime.datetime(2000, 1, 1, 12, 0)
    return dt.days + (dt.seconds + dt.microseconds / 1000000.0) / (24 * 3600.0)


def julian_centuries(model_time):
    return days_from_2000(model_time) / 36525.0


def obliquity(t):
    return np.deg2rad(23.0 + 26.0 / 60 + 21.406 / 3600.0
                      - (46.836769 * t - 0.0001831 * (t ** 2) + 0.00200340 * (t ** 3) - 0.576e-6 * (t ** 4) - 4.34e-8 * (t ** 5)) / 3600.0)


def sun_ecliptic_longitude(t):
    mean_anomaly = np.deg2rad(357.52910 + 35999.05030 * t - 0.0001559 * t * t - 0.00000048 * t * t * t)
    mean_longitude = np.deg2rad(280.46645 + 36000.76983 * t + 0.0003032 * (t ** 2))
    d_l = np.deg2rad((1.914600 - 0.004817 * t - 0.000014 * (t ** 2)) * np.sin(mean_anomaly)
                     + (0.019993 - 0.000101 * t) * np.sin(2 * mean_anomaly) + 0.000290 * np.sin(3 * mean_anomaly))
    return mean_longitude + d_l


def gmst(t):
    theta = 67310.54841 + t * (876600 * 3600 + 8640184.812866 + t * (0.093104 - t * 6.2 * 10e-6))
    theta_radians = np.deg2rad(theta / 240.0) % (2.0 * np.pi)
    if theta_radians < 0:
        theta_radians += 2.0 * np.pi
    return theta_radians


def sun_position(t):
    """-> (declination, right ascension, Greenwich mean sidereal time), all scalars (component.py:90-103)."""
    eps, eclon = obliquity(t), sun_ecliptic_longitude(t)
    x = np.cos(eclon)
    y = np.cos(eps) * np.sin(eclon)
    z = np.sin(eps) * np.sin(eclon)
    r = np.sqrt(1.0 - z * z)
    return np.arctan2(z, r), 2.0 * np.arctan2(y, (x + r)), gmst(t)


def zenith_angle(lat_deg, lon_deg, model_time):
    lat_deg, lon_deg = np.asarray(lat_deg, dtype=np.float64), np.asarray(lon_deg, dtype=np.float64)
    dec, ra, g = sun_position(julian_centuries(model_time))
    sin_lat, cos_lat = np.sin(np.deg2rad(lat_deg)), np.cos(np.deg2rad(lat_deg))
    h_angle = g + lon_deg * (np.pi / 180.0) - ra
    cos_mu = np.clip(sin_lat * np.sin(dec) + cos_lat * np.cos(dec) * np.cos(h_angle), -1.0, 1.0)
    return np.clip(np.arccos(cos_mu), -np.pi / 2.0, np.pi / 2.0)


def default_grid(nx, ny):
    """(latitude, longitude) of climt.get_grid(nx, ny) -- gaussian latitudes; a column is (0, 0)."""
    if nx is None or ny is None:
        return np.zeros((1, 1)), np.zeros((1, 1))
    x, _ = leggauss(ny)
    lat = -np.rad2deg(np.arcsin(x))
    lon = np.linspace(0.0, 360.0, nx * 2, endpoint=False)[:-1:2]
    return np.repeat(lat[:, None], nx, axis=1), np.repeat(lon[None, :], ny, axis=0)


DEFAULT_TIME = datetime.datetime(2000, 1, 1)
