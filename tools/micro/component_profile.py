"""cProfile of the drop-in component call on a host state (development tool, GPU)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RRTMG_HIP_ALLOW_SYNTHETIC_LW"] = "1"
import climt_amd
sw, lw = climt_amd.RRTMGShortwave(), climt_amd.RRTMGLongwave()
state = climt_amd.get_default_state([sw, lw], grid_state=climt_amd.get_grid(nx=128, ny=64, nz=60))
sw(state); lw(state)
t0 = time.perf_counter()
for _ in range(5): sw(state); lw(state)
print("sw+lw per call %.1f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): sw(state); lw(state)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
