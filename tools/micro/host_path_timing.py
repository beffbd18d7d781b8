"""Where the host-pointer API spends its time (development tool, GPU)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from climt_amd._lib import Context, SW_OUT, LW_OUT
from climt_amd.synthetic import make_columns
K = dict(pi=np.pi, grav=9.80665, planck=6.62607004e-27, boltz=1.38064852e-16, clight=2.99792458e10, avogad=6.022140857e23, alosmt=2.6867774e19, gascon=8.3144598e7, sbcnst=5.670367e-12, secdy=86400.0)
ctx = Context(0); ctx.set_constants(**K); ctx.sw_init(1004.64); ctx.lw_init(1004.64)
N, L = 8192, 60
c = make_columns(N, L); c.pop("lat"); c.update(icld=1, iaer=0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1)
def T(f, n=10):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("fresh outputs      sw %.2f ms  lw %.2f ms" % (T(lambda: ctx.sw_fluxes(c)), T(lambda: ctx.lw_fluxes(c))))
so = {k: np.ones((L + lev, N)) for k, lev in SW_OUT}; lo = {k: np.ones((L + lev, N)) for k, lev in LW_OUT}
print("reused outputs     sw %.2f ms  lw %.2f ms" % (T(lambda: ctx.sw_fluxes(c, out=so)), T(lambda: ctx.lw_fluxes(c, out=lo))))
print("np.zeros of the 6 outputs %.2f ms" % T(lambda: {k: np.zeros((L + lev, N)) for k, lev in SW_OUT}))
# the component path's extra arrays
big = dict(c)
big.update(taucld=np.zeros((L, N, 14)), ssacld=np.ones((L, N, 14)), asmcld=np.zeros((L, N, 14)), fsfcld=np.zeros((L, N, 14)),
           tauaer=np.zeros((14, L, N)), ssaaer=np.ones((14, L, N)), asmaer=np.zeros((14, L, N)), ecaer=np.zeros((6, L, N)))
print("sw + optional arrays, reused outputs %.2f ms" % T(lambda: ctx.sw_fluxes(big, out=so)))
bigl = dict(c); bigl.update(taucld=np.zeros((L, N, 16)), tauaer=np.zeros((16, L, N)))
print("lw + optional arrays, reused outputs %.2f ms" % T(lambda: ctx.lw_fluxes(bigl, out=lo)))
