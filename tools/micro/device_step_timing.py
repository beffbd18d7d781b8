"""Time of one model step with the state resident on the device (development tool, GPU): Instellation -> RRTMG SW + LW ->
Adams-Bashforth -> SlabSurface on a 128 x 64 x 60 grid (8192 columns), radiation refreshed EVERY step; against the same loop on
the host state."""
import os, sys, time
from datetime import timedelta
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import climt_amd
from climt_amd import _hip


def loop(device, steps, mcica=False, wait=False, rng="kissvec"):
    kw = dict(mcica=True, random_number_generator=rng) if mcica else {}
    sun, slab = climt_amd.Instellation(), climt_amd.SlabSurface()
    lw, sw = climt_amd.RRTMGLongwave(allow_synthetic_tables=True, **kw), climt_amd.RRTMGShortwave(**kw)
    state = climt_amd.get_default_state([sun, lw, sw, slab], grid_state=climt_amd.get_grid(nx=128, ny=64, nz=60))
    dt = timedelta(seconds=600)
    if device:
        state = climt_amd.DeviceState.from_host(state, [sun, lw, sw, slab])
        stepper = climt_amd.DeviceAdamsBashforth(lw, sw, slab, wait_every_step=wait)
    else:
        stepper = climt_amd.AdamsBashforth(lw, sw, slab)

    def one():
        nonlocal state
        state.update(sun(state))
        diag, state = stepper(state, dt)
        state.update(diag)
        state["time"] = state["time"] + dt
    for _ in range(3):
        one()
    _hip.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    _hip.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


if __name__ == "__main__":
    import logging; logging.disable(logging.WARNING)
    for mc in (False, True):
        d = loop(True, 200, mc)
        dw = loop(True, 200, mc, wait=True)
        h = loop(False, 5, mc)
        print("mcica=%d  device-resident %.3f ms/step (%.3g columns/s; %.3f ms with a host wait per step)   host state %.1f ms/step" % (mc, d, 8192 / (d * 1e-3), dw, h))
    d = loop(True, 200, True, rng="mersenne_twister")
    print("mcica=1 with the reference's default generator (mersenne_twister): device-resident %.3f ms/step" % d)
    if len(sys.argv) > 1 and sys.argv[1] == "profile":
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        loop(True, 200, True)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
