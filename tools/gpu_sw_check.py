import sys, time, numpy as np
sys.path.insert(0, '.')
from climt_amd._lib import Context
from climt_amd import _hip
from climt_amd.synthetic import make_columns, overcast
from oracle.ref_driver import RefSW, CONSTANTS, CPDAIR
ctx = Context(0); ctx.set_constants(**CONSTANTS); ctx.sw_init(CPDAIR)
ref = RefSW()
base = dict(icld=1, iaer=0, adjes=1.0, dyofyr=1, scon=1367.0, isolvar=0, inflg=2, iceflg=1, liqflg=1)
def cmp(name, c, mcica=False):
    r = ref.fluxes(c, mcica=mcica); g = ctx.sw_fluxes(c, mcica=mcica)
    print(name, ' '.join('%s %.2e' % (k, np.abs(g[k]-r[k]).max()) for k in g), flush=True)
c = make_columns(200, 60); c.update(base); cmp('clear', c)
c = overcast(make_columns(200, 60, cloudy=True)); c.update(base); cmp('overcast', c)
c = make_columns(200, 60, cloudy=True); c.update(base); c.update(irng=0, permuteseed=684); cmp('mcica kiss', c, True)
c.update(irng=1, permuteseed=209652396, icld=2); cmp('mcica mt maxrand', c, True)
# timing, device resident
for N, cloudy in ((8192, False), (8192, True)):
    c = make_columns(N, 60, cloudy=cloudy); c.update(base); c.update(irng=0, permuteseed=684)
    L = 60
    dev = {k: _hip.DeviceArray.from_host(v) for k, v in c.items() if isinstance(v, np.ndarray) and k != 'lat'}
    inp = {k: v.ptr for k, v in dev.items()}; inp.update({k: v for k, v in c.items() if not isinstance(v, np.ndarray)})
    inp.update(ncol=N, nlay=L)
    out = {k: _hip.DeviceArray((L + lev, N)) for k, lev in (("swuflx",1),("swdflx",1),("swhr",0),("swuflxc",1),("swdflxc",1),("swhrc",0))}
    outp = {k: v.ptr for k, v in out.items()}
    for it in range(2): ctx.sw_fluxes(inp, mcica=cloudy, out=outp, memspace=1)
    t = time.time(); n = 5
    for it in range(n): ctx.sw_fluxes(inp, mcica=cloudy, out=outp, memspace=1)
    dt = (time.time() - t) / n
    print('SW N=%d cloudy=%s: %.2f ms/call -> %.0f col/s' % (N, cloudy, dt*1e3, N/dt), flush=True)
    h = ctx.sw_fluxes(c, mcica=cloudy)
    print('  device-resident vs host-path max diff', max(np.abs(out[k].download()-h[k]).max() for k in h))
