"""McICA drop-in call with either random number generator (development tool, GPU): time per SW+LW call and where it goes."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ.setdefault("RRTMG_HIP_ALLOW_SYNTHETIC_LW", "1")
import climt_amd
for rng in ("kissvec", "mersenne_twister"):
    sw = climt_amd.RRTMGShortwave(mcica=True, cloud_overlap_method="maximum_random", random_number_generator=rng)
    lw = climt_amd.RRTMGLongwave(mcica=True, cloud_overlap_method="maximum_random", random_number_generator=rng)
    state = climt_amd.get_default_state([sw, lw], grid_state=climt_amd.get_grid(nx=128, ny=64, nz=60))
    cf = state["cloud_area_fraction_in_atmosphere_layer"].values
    cf[20:30] = 0.4
    state["mass_content_of_cloud_liquid_water_in_atmosphere_layer"].values[20:30] = 0.03
    for _ in range(3): r = (sw(state), lw(state))
    t = []
    for _ in range(10):
        t0 = time.perf_counter(); r = (sw(state), lw(state)); t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    print("%-18s McICA drop-in call: %.2f ms median (min %.2f) = %.3g columns/s" % (rng, np.median(t), t.min(), 8192 / np.median(t) * 1e3))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): r = (sw(state), lw(state))
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(5)
