#!/usr/bin/env python3
"""BASELINE.json configs[0] -- the reference's examples/radiative_equilibrium_rrtmg.py:43-66 with `from climt_amd import ...`:
one column of 30 levels relaxing towards radiative equilibrium under RRTMG shortwave + longwave, stepped with
AdamsBashforth([rad_sw, rad_lw]) (list form) at dt = 3 h.  The loop below is the reference's, line for line; the plotting
monitor (matplotlib) is replaced by a printed line.

    python examples/radiative_equilibrium.py [--steps 40] [--device-resident]

--device-resident keeps the state in HBM (climt_amd.DeviceState + DeviceAdamsBashforth): same components, same numbers.
The longwave k-distribution tables of this build are synthetic while the reference's data file is missing (the component says so).
"""
import argparse
import os
import sys
from datetime import timedelta

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climt_amd import AdamsBashforth, RRTMGLongwave, RRTMGShortwave, get_default_state, get_grid  # noqa: E402


def run(steps=40, device_resident=False, report=None):
    """-> (diagnostics of step 0, state after `steps` steps), both as host DataArrays."""
    rad_sw = RRTMGShortwave()
    rad_lw = RRTMGLongwave(allow_synthetic_tables=True)
    time_stepper = AdamsBashforth([rad_sw, rad_lw])
    timestep = timedelta(hours=3)

    grid = get_grid(nx=1, ny=1, nz=30)
    state = get_default_state([rad_sw, rad_lw], grid_state=grid)
    to_host = lambda s, names=None: {k: s[k] for k in (names or s)}
    if device_resident:
        import climt_amd
        state = climt_amd.DeviceState.from_host(state, [rad_sw, rad_lw])
        time_stepper = climt_amd.DeviceAdamsBashforth([rad_sw, rad_lw])
        to_host = lambda s, names=None: {k: s.download(k) for k in (names or s) if k != "time"}

    first = None
    for i in range(steps):

        diagnostics, new_state = time_stepper(state, timestep)
        state.update(diagnostics)
        if i == 0:
            first = to_host(state, list(diagnostics))
        if i % 2 == 0 and report is not None:
            report(i, to_host(state))
        state = new_state
    return first, to_host(state)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--device-resident", action="store_true")
    a = ap.parse_args()

    def report(i, state):
        t = state["air_temperature"].values.ravel()
        print("step %4d  T(surface layer) %8.3f K  T(top) %8.3f K  OLR %8.3f  SW heating(top) %7.3f K/day  LW heating(top) %8.3f K/day" % (
            i, t[0], t[-1], state["upwelling_longwave_flux_in_air"].values.ravel()[-1],
            state["air_temperature_tendency_from_shortwave"].values.ravel()[-1],
            state["air_temperature_tendency_from_longwave"].values.ravel()[-1]))
    run(a.steps, a.device_resident, report)


if __name__ == "__main__":
    main()
