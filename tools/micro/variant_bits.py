"""Do the two shortwave solve variants give a cloud-free column the same bits? (development tool, GPU)
256 cloud-free columns through the clear-sky variant, then through the cloudy variant (one cloudy column put into each tile).
  RRTMG_HIP_LIB=<variant .so> python tools/micro/variant_bits.py        (docs/EXPERIMENTS.md, round 4)"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from climt_amd._lib import Context
from climt_amd.synthetic import make_columns
K = dict(pi=np.pi, grav=9.80665, planck=6.62607004e-27, boltz=1.38064852e-16, clight=2.99792458e10, avogad=6.022140857e23, alosmt=2.6867774e19, gascon=8.3144598e7, sbcnst=5.670367e-12, secdy=86400.0)
ctx = Context(0); ctx.set_constants(**K); ctx.sw_init(1004.64); ctx.lw_init(1004.64)
N, L = 256, 60
c = make_columns(N, L, cloudy=False, seed=3); c.pop("lat")
c.update(icld=1, iaer=0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1, irng=0, permuteseed=684)
clear = ctx.sw_fluxes(c, mcica=True)            # all tiles cloud-free: the clear-sky variant
d = dict(c); d["cldfr"] = c["cldfr"].copy(); d["cliqwp"] = c["cliqwp"].copy()
for t in range(N // 64):                        # one cloudy column per tile: the other 63 run the cloudy variant
    d["cldfr"][20:24, 64 * t] = 1.0; d["cliqwp"][20:24, 64 * t] = 40.0
mixed = ctx.sw_fluxes(d, mcica=True)
keep = np.ones(N, bool); keep[::64] = False
worst = max(float(np.abs(clear[k][:, keep] - mixed[k][:, keep]).max()) for k in clear)
same = all(np.array_equal(clear[k][:, keep], mixed[k][:, keep]) for k in clear)
print(os.environ.get("RRTMG_HIP_LIB", "product"), "cloud-free columns, clear variant vs cloudy variant: bitwise", same, "max |d|", worst)
