"""Deterministic synthetic column sets for the benchmark and the parity tests
(config 2 = clear sky, config 3 = liquid+ice cloud with McICA).

Everything is produced at the C-ABI boundary of the RRTMG path (after sympl's unit conversion):
pressures in hPa (mbar), temperatures in K, volume mixing ratios, water paths in g m^-2, sizes in
micron, arrays C-contiguous [layer, column] with layer 0 at the surface.
"""
import os

import numpy as np

SEED = 20260927


def _hybrid_interfaces(ps_hpa, nlay, ptop_hpa=0.2):
    """Interface pressures [nlay+1, ncol]: sigma-like near the surface relaxing to pure pressure aloft."""
    k = np.arange(nlay + 1) / nlay
    # eta runs 1 -> 0 with finer spacing near the surface and in the stratosphere
    eta = (1.0 - k) ** 1.6
    pref = ptop_hpa + (1013.2 - ptop_hpa) * eta
    b = np.clip((pref - 150.0) / (1013.2 - 150.0), 0.0, 1.0) ** 1.2
    a = pref - b * 1013.2
    return a[:, None] + b[:, None] * ps_hpa[None, :]


def _qsat(t, p_hpa):
    es = 6.112 * np.exp(17.67 * (t - 273.15) / (t - 29.65))
    es = np.minimum(es, 0.5 * p_hpa)
    return 0.622 * es / (p_hpa - 0.378 * es)


_OZONE = None


def _ozone_profile(p_pa):
    global _OZONE
    from .initialization import not_a_knot_spline
    if _OZONE is None:
        tab = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "ozone_profile.npz"))
        _OZONE = (tab["pressure_Pa"], tab["mole_fraction"])
    return not_a_knot_spline(p_pa, _OZONE[0], _OZONE[1])


def make_columns(ncol, nlay=60, cloudy=False, seed=SEED, nlat=None):
    """Return a dict of boundary-level inputs shared by the LW and SW entry points."""
    rng = np.random.default_rng(seed)
    if nlat is None:
        nlat = max(1, int(round(np.sqrt(ncol / 2.0))))
    lat_nodes = np.arcsin(np.polynomial.legendre.leggauss(nlat)[0])
    lat = lat_nodes[np.arange(ncol) % nlat]
    u1 = rng.uniform(-1, 1, ncol)
    u2 = rng.uniform(-1, 1, ncol)
    ps = 1013.2 * (1.0 + 0.02 * u1)
    tsfc = 290.0 - 40.0 * np.sin(lat) ** 2 + 2.0 * u2
    plev = _hybrid_interfaces(ps, nlay)
    play = 0.5 * (plev[:-1] + plev[1:])
    # temperature: 6.5 K/km lapse to a tropopause, then +1 K/km (log-pressure height, H = 7.5 km)
    z = 7.5 * np.log(ps[None, :] / play)
    ttrop = 200.0 + 15.0 * np.abs(np.sin(lat))[None, :]
    ztrop = (tsfc[None, :] - 1.5 - ttrop) / 6.5
    tlay = np.where(z < ztrop, tsfc[None, :] - 1.5 - 6.5 * z, ttrop + 1.0 * (z - ztrop))
    tlay = tlay + 0.3 * rng.standard_normal(tlay.shape)
    # interface temperatures exactly as the climt host computes them (_core/util.py:125-142)
    lp = np.log(play)
    tlev = np.zeros((nlay + 1, ncol))
    w = (np.log(plev[1:-1]) - lp[1:]) / (lp[:-1] - lp[1:])
    tlev[1:-1] = tlay[1:] - w * (tlay[1:] - tlay[:-1])
    tlev[0] = tsfc
    tlev[-1] = tlay[-1]
    q = np.maximum(0.7 * _qsat(tlay, play), 3.0e-6)
    h2o = q * 28.964 / 18.02
    # ozone: climt's default profile -- the not-a-knot spline of its 30-point table, evaluated at the layer pressures
    # (climt/_core/initialization.py:1130-1141); outside the table the end cubics continue, as in climt
    o3 = _ozone_profile(100.0 * play)
    full = lambda v: np.full((nlay, ncol), v)
    inp = dict(
        play=play, plev=plev, tlay=tlay, tlev=tlev, tsfc=tsfc, h2o=h2o, o3=o3,
        co2=full(330e-6), ch4=full(1.7e-6), n2o=full(0.3e-6), o2=full(0.21),
        cfc11=full(0.25e-9), cfc12=full(0.5e-9), cfc22=full(0.1e-9), ccl4=full(0.1e-9),
        emis=np.ones((16, ncol)),
        asdir=np.full(ncol, 0.06) + 0.02 * u1 ** 2, asdif=np.full(ncol, 0.06) + 0.02 * u2 ** 2,
        aldir=np.full(ncol, 0.06) + 0.03 * u1 ** 2, aldif=np.full(ncol, 0.06) + 0.03 * u2 ** 2,
        coszen=np.cos(np.abs(lat)),
        lat=lat,
    )
    cld = np.zeros((nlay, ncol))
    clwp = np.zeros((nlay, ncol))
    ciwp = np.zeros((nlay, ncol))
    if cloudy:
        region = (np.arange(ncol) // max(1, nlat // 4)) % 4
        frac = np.array([0.0, 0.3, 0.6, 1.0])[region]
        inband = (play > 300.0) & (play < 850.0)
        # broken cloud decks: three-layer slabs separated by clear layers
        deck = ((np.arange(nlay)[:, None] + region[None, :]) % 5) < 3
        cld = np.where(inband & deck, frac[None, :], 0.0)
        clwp = np.where((cld > 0) & (tlay > 253.0), rng.uniform(20.0, 80.0, cld.shape), 0.0)
        ciwp = np.where((cld > 0) & (tlay < 263.0), rng.uniform(5.0, 30.0, cld.shape), 0.0)
        cld = np.where((clwp + ciwp) > 0, cld, 0.0)
    inp.update(cldfr=cld, cliqwp=clwp, cicewp=ciwp, reliq=full(10.0), reice=full(30.0))
    return {k: (np.ascontiguousarray(v, dtype=np.float64) if isinstance(v, np.ndarray) else v) for k, v in inp.items()}


def overcast(inp):
    """Variant for the non-McICA shortwave path, which accepts only clear or overcast layers."""
    out = dict(inp)
    out["cldfr"] = np.where(inp["cldfr"] > 0.5, 1.0, 0.0)
    keep = out["cldfr"] > 0
    out["cliqwp"] = np.where(keep, inp["cliqwp"], 0.0)
    out["cicewp"] = np.where(keep, inp["cicewp"], 0.0)
    return out
