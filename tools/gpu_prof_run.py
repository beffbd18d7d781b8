"""Minimal device-resident LW+SW loop for rocprofv3 (development tool)."""
import sys, numpy as np
sys.path.insert(0, '.')
from climt_amd._lib import Context, SW_OUT, LW_OUT
from climt_amd import _hip
from climt_amd.synthetic import make_columns
CONSTANTS = dict(pi=np.pi, grav=9.80665, planck=6.62607004e-27, boltz=1.38064852e-16, clight=2.99792458e10,
                 avogad=6.022140857e23, alosmt=2.6867774e19, gascon=8.3144598e7, sbcnst=5.670367e-12, secdy=86400.0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cloudy = len(sys.argv) > 2 and sys.argv[2] == 'cloudy'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = Context(0); ctx.set_constants(**CONSTANTS); ctx.sw_init(1004.64); ctx.lw_init(1004.64)
c = make_columns(N, 60, cloudy=cloudy); c.update(icld=1, iaer=0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1, irng=0, permuteseed=684)
L = 60
dev = {k: _hip.DeviceArray.from_host(v) for k, v in c.items() if isinstance(v, np.ndarray) and k != 'lat'}
inp = {k: v.ptr for k, v in dev.items()}; inp.update({k: v for k, v in c.items() if not isinstance(v, np.ndarray)}); inp.update(ncol=N, nlay=L)
so = {k: _hip.DeviceArray((L + lev, N)) for k, lev in SW_OUT}; lo = {k: _hip.DeviceArray((L + lev, N)) for k, lev in LW_OUT}
sop = {k: v.ptr for k, v in so.items()}; lop = {k: v.ptr for k, v in lo.items()}
for it in range(steps):
    ctx.sw_fluxes(inp, mcica=cloudy, out=sop, memspace=1); ctx.lw_fluxes(inp, mcica=cloudy, out=lop, memspace=1)
print('done')
