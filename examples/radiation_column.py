#!/usr/bin/env python3
"""A radiative column on one MI355X with nothing but this package: grid + default state, zenith angle from
Instellation, RRTMG longwave + shortwave refreshed every hour of model time, a slab ocean underneath, stepped with
Adams-Bashforth -- the radiation part of the reference's examples/gmd_aquaplanet.py:61-104 (its dynamical core,
convection and boundary layer are out of scope here).

    python examples/radiation_column.py [--nx 32 --ny 16 --nz 28 --hours 6] [--device-resident]

--device-resident keeps the model state in HBM (climt_amd.DeviceState): it is uploaded once, the same component
instances run their device paths, the tendency sum and the Adams-Bashforth update are kernels, and only the printed
diagnostics come back to the host.
"""
import argparse
import os
import sys
from datetime import timedelta

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import climt_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=32)
    ap.add_argument("--ny", type=int, default=16)
    ap.add_argument("--nz", type=int, default=28)
    ap.add_argument("--hours", type=float, default=6.0)
    ap.add_argument("--dt", type=float, default=600.0, help="model time step, s")
    ap.add_argument("--device-resident", action="store_true", help="state resident on the GPU (climt_amd.DeviceState)")
    a = ap.parse_args()

    sun = climt_amd.Instellation()
    lw = climt_amd.UpdateFrequencyWrapper(climt_amd.RRTMGLongwave(allow_synthetic_tables=True), timedelta(hours=1))
    sw = climt_amd.UpdateFrequencyWrapper(climt_amd.RRTMGShortwave(), timedelta(hours=1))
    slab = climt_amd.SlabSurface()

    grid = climt_amd.get_grid(nx=a.nx, ny=a.ny, nz=a.nz)
    state = climt_amd.get_default_state([sun, lw, sw, slab], grid_state=grid)
    p = state["air_pressure"].values
    state["air_temperature"].values[:] = np.maximum(200.0, 290.0 * (p / 1.0e5) ** 0.19)
    state["specific_humidity"].values[:] = 0.012 * (p / 1.0e5) ** 3
    dt = timedelta(seconds=a.dt)
    if a.device_resident:
        state = climt_amd.DeviceState.from_host(state, [sun, lw, sw, slab])
        # (the host need not wait for a step: the diagnostics printed below are downloaded, which synchronizes)
        stepper = climt_amd.DeviceAdamsBashforth(lw, sw, slab, wait_every_step=False)
        host = lambda name: state.download(name).values
    else:
        stepper = climt_amd.AdamsBashforth(lw, sw, slab)
        host = lambda name: state[name].values
    for step in range(int(a.hours * 3600 / a.dt)):
        state.update(sun(state))
        diag, state = stepper(state, dt)
        state.update(diag)
        state["time"] = state["time"] + dt
        if step % 6 == 0:
            olr = host("upwelling_longwave_flux_in_air")[-1].mean()
            asr = (host("downwelling_shortwave_flux_in_air")[-1] - host("upwelling_shortwave_flux_in_air")[-1]).mean()
            print("%s  OLR %7.2f  absorbed solar %7.2f  Ts %7.3f  T(lowest) %7.3f" % (
                state["time"], olr, asr, host("surface_temperature").mean(), host("air_temperature")[0].mean()))


if __name__ == "__main__":
    main()
