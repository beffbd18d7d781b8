#!/usr/bin/env python3
"""Duration of every work item of the solve kernels run ALONE (development tool, GPU): the launch-order position k of
RRTMG_HIP_ONLY_ITEM=k is the only one that computes, so kernel_ms is that item's duration for all tiles.  Needs a PROFILE
build of the library (the product build has no such switch, climt_amd/csrc/rrtmg_profile.h):
    RRTMG_HIP_BUILD_FLAGS=-DRRTMG_PROFILE RRTMG_HIP_BUILD_OUT=$PWD/climt_amd/_lib/lib_profile.so python climt_amd/build.py --force
    RRTMG_HIP_LIB=$PWD/climt_amd/_lib/lib_profile.so python tools/item_times.py [ncol=8192] [cloudy]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climt_amd import _hip
from climt_amd._lib import LW_OUT, SW_OUT, Context
from climt_amd.synthetic import make_columns
CONSTANTS = dict(pi=np.pi, grav=9.80665, planck=6.62607004e-27, boltz=1.38064852e-16, clight=2.99792458e10,
                 avogad=6.022140857e23, alosmt=2.6867774e19, gascon=8.3144598e7, sbcnst=5.670367e-12, secdy=86400.0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cloudy = len(sys.argv) > 2 and sys.argv[2] == "cloudy"
L = 60
SW_NG = [6, 12, 8, 8, 10, 10, 2, 10, 8, 6, 6, 8, 6, 12]
SW_NSPA = [9, 9, 9, 9, 1, 9, 9, 1, 9, 1, 0, 1, 9, 1]
LW_NG = [10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2]
LW_NSPA = [1, 1, 9, 9, 9, 1, 9, 1, 9, 1, 1, 9, 9, 1, 9, 9]


def order(ngs, nspa, base, fac, gmax, hi, lo):
    items = []
    for b, ng in enumerate(ngs):
        ig = 0
        while ig < ng:
            g = 4 if (gmax == 4 and ng - ig >= 4) else 2
            items.append((b + base, ig, g, (hi if nspa[b] == 9 else lo) + g * fac))
            ig += g
    idx = sorted(range(len(items)), key=lambda i: -items[i][3])       # python's sort is stable, as the insertion sort
    return [items[i] for i in idx]


ctx = Context(0); ctx.set_constants(**CONSTANTS); ctx.sw_init(1004.64); ctx.lw_init(1004.64)
c = make_columns(N, L, cloudy=cloudy); c.pop("lat")
c.update(icld=1, iaer=0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1, irng=0, permuteseed=684)
dev = {k: _hip.DeviceArray.from_host(v) for k, v in c.items() if isinstance(v, np.ndarray)}
inp = {k: v.ptr for k, v in dev.items()}; inp.update({k: v for k, v in c.items() if not isinstance(v, np.ndarray)}); inp.update(ncol=N, nlay=L)
so = {k: _hip.DeviceArray((L + lev, N)) for k, lev in SW_OUT}; lo = {k: _hip.DeviceArray((L + lev, N)) for k, lev in LW_OUT}
sop = {k: v.ptr for k, v in so.items()}; lop = {k: v.ptr for k, v in lo.items()}


def run(which, k):
    if k is None:
        os.environ.pop("RRTMG_HIP_ONLY_ITEM", None)
    else:
        os.environ["RRTMG_HIP_ONLY_ITEM"] = str(k)
    t = []
    for _ in range(4):
        if which == "sw":
            ctx.sw_fluxes(inp, mcica=cloudy, out=sop, memspace=1)
        else:
            ctx.lw_fluxes(inp, mcica=cloudy, out=lop, memspace=1)
        t.append(ctx.kernel_ms(which, cloudy=cloudy))
    return float(np.median(t[1:]))


for which, its in (("sw", order(SW_NG, SW_NSPA, 16, 1.0, 2 if cloudy else 4, 1.0, 0.6)), ("lw", order(LW_NG, LW_NSPA, 1, 0.7, 4, 2.0, 1.0))):
    full = run(which, None)
    ts = [run(which, k) for k in range(len(its))]
    print("%s %s N=%d: whole kernel %.3f ms; items alone: sum %.3f ms, max %.3f, mean %.3f" % (which, "cloudy" if cloudy else "clear", N, full, sum(ts), max(ts), np.mean(ts)))
    for k, ((band, ig, g, cost), t) in enumerate(zip(its, ts)):
        print("  k=%2d band %2d ig0 %2d G=%d model %.1f  alone %.3f ms" % (k, band, ig, g, cost, t))
