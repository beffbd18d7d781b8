"""Where a drop-in component call spends its time (development tool, GPU): RRTMGShortwave()(state) + RRTMGLongwave()(state) on
get_default_state(128 x 64 x 60), with the input products (unit conversions, water-vapour mixing ratio) formed in the background
into kept buffers (climt_amd.rrtmg.common.InputStaging) or one after the other into fresh arrays as the reference does."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("RRTMG_HIP_ALLOW_SYNTHETIC_LW", "1")
import climt_amd
from climt_amd.rrtmg import common

def run(tag, n=12):
    sw, lw = climt_amd.RRTMGShortwave(), climt_amd.RRTMGLongwave()
    state = climt_amd.get_default_state([sw, lw], grid_state=climt_amd.get_grid(nx=128, ny=64, nz=60))
    for _ in range(3): r = (sw(state), lw(state))
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); r = (sw(state), lw(state)); t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    print("%-28s %.2f ms per SW+LW call (min %.2f)  = %.3g columns/s" % (tag, np.median(t), t.min(), 8192 / np.median(t) * 1e3))
    return r

a = run("staged, concurrent")
def fresh(self, name, values, factor, divisor=None, pieces=1):
    out = values * factor
    return out / divisor if divisor is not None else out
common.InputStaging.scaled = fresh
b = run("fresh arrays, sequential")
same = all(np.array_equal(x[k].values, y[k].values) for x, y in zip(a[0] + a[1], b[0] + b[1]) for k in x)
print("identical results:", same)
